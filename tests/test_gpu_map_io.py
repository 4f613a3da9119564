"""-m gpu: the on-disk map formats as the loop uses them — a map that was OPTIMISED on the device goes to
`params.npz` and comes back, and what comes back must render to the same bits; the semantic PLY of that map is checked attribute by
attribute at the byte offsets the reference's exporter defines (scripts/export_ply_semantic_tree.py:279-327: one binary little-endian
`vertex` element — x, y, z, nx, ny, nz as f4, red, green, blue as u1, opacity, scale_0..2, rot_0..3 as f4: 59 bytes per Gaussian).
`plyfile` (what the reference writes PLY through) is not importable here, so the layout is pinned by those offsets, not by that package:
parity unpinned in that sense.  The CPU suite (tests/test_map_io.py) covers the schema and the header."""
import os
import struct

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mapped_scene(iters=3):
    """a few mapping iterations (Adam on every Gaussian parameter, as scripts/hierslam.py:2016-2057) so that what is saved is a map the
    device produced, not the generator's arrays"""
    from test_gpu_slam_loop import _render, _scene
    from hsr_utils import losses as L
    cam, params, (W, H, K) = _scene(P=8000, W=256, H=192, K=12)
    params["cam_unnorm_rots"] = torch.tensor([1.0, 0, 0, 0]).view(1, 4, 1).cuda()
    params["cam_trans"] = torch.zeros(1, 3, 1).cuda()
    with torch.no_grad():
        im_gt, _, sem_gt, depth_gt, _, _ = _render(params, cam, 0, False, False)
    g = torch.Generator().manual_seed(5)
    keys = ("means3D", "rgb_colors", "semantic", "logit_opacities", "log_scales", "unnorm_rotations")
    for k in keys:
        params[k] = (params[k] + 0.03 * torch.randn(params[k].shape, generator=g).cuda() * params[k].abs().mean()).requires_grad_(True)
    opt = torch.optim.Adam([params[k] for k in keys], lr=1e-3)
    for _ in range(iters):
        im, radius, sem, depth, med, opac = _render(params, cam, 0, True, False)
        loss = L.masked_l1(depth, depth_gt, (depth_gt > 0), "sum") + 0.5 * (im - im_gt).abs().sum() + 0.01 * (sem - sem_gt).abs().sum()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    return cam, {k: v.detach() for k, v in params.items()}, (W, H, K)


def test_saved_map_reloads_and_renders_bit_identically(tmp_path):
    from test_gpu_slam_loop import _render
    from hsr_utils import map_io as M
    cam, params, (W, H, K) = _mapped_scene()
    with torch.no_grad():
        before = _render(params, cam, 0, False, False)
    variables = {"timestep": torch.zeros(params["means3D"].shape[0]).cuda()}
    full = M.finalize_params(params, variables, torch.eye(3), torch.eye(4), W, H, [torch.eye(4)], [0])
    path = M.save_params(full, str(tmp_path))
    back = M.load_params(path, device="cuda")
    assert M.check_schema({k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in back.items()})[0] == params["means3D"].shape[0]
    for k in M.GAUSSIAN_KEYS + M.CAMERA_KEYS:
        assert torch.equal(back[k], params[k]), k          # fp32 in, fp32 out: every bit
    with torch.no_grad():
        after = _render({k: back[k] for k in M.GAUSSIAN_KEYS + M.CAMERA_KEYS}, cam, 0, False, False)
    names = ("color", "radii", "semantic", "depth", "median depth", "opacity")
    for n, a, b in zip(names, before, after):
        assert torch.equal(a, b), "%s of the reloaded map differs" % n
    assert int((before[1] > 0).sum()) > 1000


def test_semantic_ply_attributes_at_the_reference_byte_offsets(tmp_path):
    from hsr_utils import map_io as M
    cam, params, (W, H, K) = _mapped_scene(iters=2)
    P = params["means3D"].shape[0]
    means = params["means3D"].cpu().numpy()
    scales = params["log_scales"].cpu().numpy()             # [P, 1]: isotropic, tiled to three columns by the exporter
    rots = params["unnorm_rotations"].cpu().numpy()
    opac = params["logit_opacities"].cpu().numpy()
    # per-Gaussian colour from the tree labels, as the exporter does (transfer_tree_label -> colour table lookup)
    labels = M.transfer_tree_label(params["semantic"].cpu().numpy(), [4, 8, 20])
    table = (np.arange(8 * 3).reshape(8, 3) * 29 % 256).astype(np.uint8)
    colors = table[labels[1]]
    path = M.save_ply_semantic(str(tmp_path / "map_semantic.ply"), means, scales, rots, colors, opac)
    raw = open(path, "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    header = raw[:end].decode("ascii").split("\n")
    assert header[0] == "ply" and header[1] == "format binary_little_endian 1.0" and header[2] == "element vertex %d" % P
    props = [tuple(l.split()[1:]) for l in header if l.startswith("property")]
    assert props == [("float", n) for n in ("x", "y", "z", "nx", "ny", "nz")] + [("uchar", n) for n in ("red", "green", "blue")] + \
        [("float", n) for n in ("opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3")]
    body = raw[end:]
    STRIDE = 6 * 4 + 3 + 8 * 4
    assert len(body) == P * STRIDE == P * 59
    rec = np.frombuffer(body, dtype=np.uint8).reshape(P, STRIDE)
    f4 = lambda off: np.ascontiguousarray(rec[:, off:off + 4]).view("<f4").reshape(P)
    for i in range(3):
        assert np.array_equal(f4(4 * i), means[:, i]), "xyz"[i]
        assert not f4(12 + 4 * i).any()                                          # normals: zeros
        assert np.array_equal(rec[:, 24 + i], colors[:, i]), ("red", "green", "blue")[i]
        assert np.array_equal(f4(31 + 4 * i), scales[:, 0]), "scale_%d" % i      # one log-scale, three columns
    assert np.array_equal(f4(27), opac[:, 0])
    for i in range(4):
        assert np.array_equal(f4(43 + 4 * i), rots[:, i]), "rot_%d" % i
    # one record by hand, through struct (no numpy views): the first Gaussian
    x, y, z = struct.unpack_from("<3f", body, 0)
    r, g_, b = struct.unpack_from("<3B", body, 24)
    assert (x, y, z) == tuple(float(v) for v in means[0]) and (r, g_, b) == tuple(int(v) for v in colors[0])
    # and the reader returns the same table
    el = M.read_ply(path)
    assert np.array_equal(el["rot_3"], rots[:, 3]) and np.array_equal(el["green"], colors[:, 1]) and el.dtype.itemsize == 59
    # the plain (SH band 0) export of the same map: 17 floats per Gaussian
    path2 = M.save_ply(str(tmp_path / "map.ply"), means, scales, rots, params["rgb_colors"].cpu().numpy(), opac)
    raw2 = open(path2, "rb").read()
    body2 = raw2[raw2.index(b"end_header\n") + 11:]
    assert len(body2) == P * 17 * 4
    tab = np.frombuffer(body2, dtype="<f4").reshape(P, 17)
    assert np.array_equal(tab[:, 0:3], means) and np.array_equal(tab[:, 9], opac[:, 0]) and np.array_equal(tab[:, 13:17], rots)
    assert np.allclose(tab[:, 6:9], (params["rgb_colors"].cpu().numpy() - 0.5) / M.C0, rtol=1e-6, atol=1e-7)
