"""On-disk map formats of the reference, so that maps built with this rasterizer stay loadable by the reference's
evaluation / visualisation / export scripts.  Host-side only.

    params2cpu, save_params, save_params_ckpt          utils/common_utils.py:25-52   (np.savez of every entry, `params.npz`)
    finalize_params                                    scripts/hierslam.py:2163-2175 (timestep, intrinsics, w2c, org_width/height,
                                                                                       gt_w2c_all_frames, keyframe_time_indices)
    load_params                                        what scripts/export_ply_semantic_tree.py:403 and the viz scripts do
    save_ply / save_ply_semantic                       scripts/export_ply_semantic_tree.py:251-327 (attribute names, order, dtypes)
    transfer_tree_label                                scripts/export_ply_semantic_tree.py:208-228

The reference writes PLY through the third-party `plyfile` package (binary, native byte order, one `vertex` element);
that package is not a dependency here: the same byte layout is written directly.
"""
import os

import numpy as np
import torch

C0 = 0.28209479177387814  # SH band-0 constant used by rgb_to_spherical_harmonic (export_ply_semantic_tree.py:201-202)

GAUSSIAN_KEYS = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales", "semantic")
CAMERA_KEYS = ("cam_unnorm_rots", "cam_trans")
EXTRA_KEYS = ("timestep", "intrinsics", "w2c", "org_width", "org_height", "gt_w2c_all_frames", "keyframe_time_indices")


def params2cpu(params):
    return {k: (v.detach().cpu().contiguous().numpy() if isinstance(v, torch.Tensor) else v) for k, v in params.items()}


def save_params(output_params, output_dir, name="params.npz"):
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, name)
    np.savez(path, **params2cpu(output_params))
    return path


def save_params_ckpt(output_params, output_dir, time_idx):
    return save_params(output_params, output_dir, "params" + str(time_idx) + ".npz")


def finalize_params(params, variables, intrinsics, first_frame_w2c, org_width, org_height, gt_w2c_all_frames, keyframe_time_indices):
    """the camera / bookkeeping entries the reference adds before its final save (scripts/hierslam.py:2163-2175)"""
    out = dict(params)
    out["timestep"] = variables["timestep"]
    out["intrinsics"] = intrinsics.detach().cpu().numpy() if isinstance(intrinsics, torch.Tensor) else np.asarray(intrinsics)
    out["w2c"] = first_frame_w2c.detach().cpu().numpy() if isinstance(first_frame_w2c, torch.Tensor) else np.asarray(first_frame_w2c)
    out["org_width"], out["org_height"] = org_width, org_height
    out["gt_w2c_all_frames"] = np.stack([w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w)
                                         for w in gt_w2c_all_frames], axis=0)
    out["keyframe_time_indices"] = np.array(keyframe_time_indices)
    return out


def load_params(path, device=None):
    """dict(np.load(path)) as the reference's consumers do; tensors on `device` for the float arrays if given.  Object arrays
    are refused (allow_pickle=False): the schema above has none."""
    z = np.load(path, allow_pickle=False)
    out = {k: z[k] for k in z.files}
    if device is not None:
        out = {k: (torch.tensor(v, device=device) if isinstance(v, np.ndarray) and v.dtype.kind == "f" and v.ndim > 0 else v)
               for k, v in out.items()}
    return out


def check_schema(params):
    """shapes the rasterizer-side code relies on; raises ValueError naming the first violation"""
    P = params["means3D"].shape[0]
    want = {"means3D": (P, 3), "rgb_colors": (P, 3), "unnorm_rotations": (P, 4), "logit_opacities": (P, 1)}
    for k, shp in want.items():
        if tuple(params[k].shape) != shp:
            raise ValueError("%s has shape %s, expected %s" % (k, tuple(params[k].shape), shp))
    if params["log_scales"].shape[0] != P or params["log_scales"].shape[1] not in (1, 3):
        raise ValueError("log_scales has shape %s, expected (%d, 1|3)" % (tuple(params["log_scales"].shape), P))
    if "semantic" in params and params["semantic"].shape[0] != P:
        raise ValueError("semantic has %d rows, expected %d" % (params["semantic"].shape[0], P))
    F = params["cam_unnorm_rots"].shape[-1]
    if tuple(params["cam_unnorm_rots"].shape) != (1, 4, F) or tuple(params["cam_trans"].shape) != (1, 3, F):
        raise ValueError("cam_unnorm_rots / cam_trans must be [1,4,F] / [1,3,F]")
    return P, F


_PLY_TYPES = {"f4": "float", "u1": "uchar", "i4": "int", "f8": "double", "u2": "ushort", "i2": "short", "i1": "char", "u4": "uint"}


def write_ply(path, elements):
    """one `vertex` element from a structured array, binary little endian — the bytes plyfile's
    PlyData([PlyElement.describe(elements, 'vertex')]).write(path) produces on a little-endian host"""
    elements = np.ascontiguousarray(elements)
    lines = ["ply", "format binary_little_endian 1.0", "element vertex %d" % elements.shape[0]]
    le_fields = []
    for name in elements.dtype.names:
        dt = elements.dtype.fields[name][0]
        code = dt.kind + str(dt.itemsize)
        lines.append("property %s %s" % (_PLY_TYPES[code], name))
        le_fields.append((name, "<" + code if dt.itemsize > 1 else code))
    lines.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("ascii"))
        f.write(elements.astype(np.dtype(le_fields), copy=False).tobytes())
    return path


def read_ply(path):
    """inverse of write_ply (binary little endian, scalar properties, one vertex element)"""
    inv = {v: k for k, v in _PLY_TYPES.items()}
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt = f.readline().split()
        if fmt[1] != b"binary_little_endian":
            raise ValueError("only binary_little_endian PLY is supported, got %s" % fmt[1].decode())
        n, fields = 0, []
        while True:
            tok = f.readline().split()
            if tok[0] == b"end_header":
                break
            if tok[0] == b"element":
                n = int(tok[2])
            elif tok[0] == b"property":
                fields.append((tok[2].decode(), "<" + inv[tok[1].decode()]))
        return np.frombuffer(f.read(), dtype=np.dtype(fields), count=n)


def _attributes(means, normals, colors, opacities, scales, rotations, color_fields):
    if normals is None:
        normals = np.zeros_like(means)
    if scales.shape[1] == 1:
        scales = np.tile(scales, (1, 3))
    dtype_full = [("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4")] + color_fields + \
                 [("opacity", "f4"), ("scale_0", "f4"), ("scale_1", "f4"), ("scale_2", "f4"),
                  ("rot_0", "f4"), ("rot_1", "f4"), ("rot_2", "f4"), ("rot_3", "f4")]
    el = np.empty(means.shape[0], dtype=dtype_full)
    cols = np.concatenate((means, normals, colors, opacities.reshape(-1, 1), scales, rotations), axis=1)
    for i, (name, _) in enumerate(dtype_full):
        el[name] = cols[:, i]
    return el


def save_ply(path, means, scales, rotations, rgbs, opacities, normals=None):
    """export_ply_semantic_tree.py:251-277: colour as SH band 0 in f_dc_0..2; log-scales / logit-opacities / quaternions as stored"""
    colors = (np.asarray(rgbs, np.float32) - 0.5) / C0
    el = _attributes(means, normals, colors, opacities, scales, rotations, [("f_dc_0", "f4"), ("f_dc_1", "f4"), ("f_dc_2", "f4")])
    return write_ply(path, el)


def transfer_tree_label(semantics, tree_num_semantic):
    """[N, K] per-Gaussian tree logits -> [levels, N] argmax label per level (export_ply_semantic_tree.py:208-228: the last
    entry of tree_num_semantic is the leaf-class count and is not a level of the embedding)"""
    sem = np.asarray(semantics)
    out, begin = [], 0
    for n in list(tree_num_semantic)[:-1]:
        out.append(np.argmax(sem[:, begin:begin + n], axis=1))
        begin += n
    return np.stack(out, axis=0)


def save_ply_semantic(path, means, scales, rotations, colors_u8, opacities, normals=None):
    """export_ply_semantic_tree.py:279-327 after its colour lookup: per-Gaussian uint8 colours in red/green/blue"""
    el = _attributes(means, normals, np.asarray(colors_u8).astype(np.float64), opacities, scales, rotations,
                     [("red", "u1"), ("green", "u1"), ("blue", "u1")])
    return write_ply(path, el)
