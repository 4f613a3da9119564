"""Silhouette densification of a mapped frame, with the reference's function names.

    non_presence_points(...)                 mask (scripts/hierslam.py:1271-1278, :1289-1290) + get_pointcloud(..., mask=...,
                                             compute_mean_sq_dist=True) (:144-194) in one device pipeline (include/hsr_densify.h)
    initialize_new_params_semantic(...)      scripts/hierslam.py:1137-1167
    initialize_new_params(...)               scripts/hierslam.py:1110-1135
    add_new_gaussians_semantic_newrender(...) scripts/hierslam.py:1307-1352 — the one the mapping loop calls for semantic maps (:1949)
    add_new_gaussians_newtest(...)           scripts/hierslam.py:1214-1262 — the one it calls for plain maps (:1945)
    add_new_gaussians_semantic(...)          scripts/hierslam.py:1264-1305 (depth + silhouette render; dead in the reference's loop)
each = fused input preparation + ONE no-grad render with this repo's rasterizer + non_presence_points + the concatenation.
Prune / optimizer-preserving concat: hsr_utils/slam_external.py (one fused device compaction).

The Parameter / bookkeeping concatenation is torch, as in the reference (it is bookkeeping on torch objects)."""
import ctypes as C

import numpy as np
import torch

from diff_gaussian_rasterization import _C as _glue

_lib = _glue._lib
_vp, _ci, _cf, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_lib.hsr_densify_scratch_bytes.restype = _sz
_lib.hsr_densify_scratch_bytes.argtypes = [_ci, _ci]
_lib.hsr_densify_frame.restype = _ci
_lib.hsr_densify_frame.argtypes = [_ci, _ci, _vp, _vp, _vp, _vp, _cf, _cf, _cf, _cf, _vp, _cf, _cf, _ci] + [_vp] * 7 + [_vp, _sz, _vp]


def _plane(t, H, W, what):
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError("hsr_utils.densify: %s must be a float32 tensor on a HIP device; there is no CPU path" % what)
    return t.reshape(-1, H, W).contiguous()


def non_presence_points(silhouette, render_depth, gt_depth, color, intrinsics, w2c, sil_thres, depth_factor=50.0):
    """Returns (new_pt_cld [M,6] = world xyz + rgb in row-major pixel order, mean3_sq_dist [M], non_presence_mask [H*W] bool,
    log_scales [M,1]).  One host sync (the count M), like the reference's `torch.sum(non_presence_mask) > 0`."""
    H, W = gt_depth.shape[-2:]
    dev = gt_depth.device
    sil, rd, gt = _plane(silhouette, H, W, "silhouette"), _plane(render_depth, H, W, "render_depth"), _plane(gt_depth, H, W, "gt_depth")
    col = _plane(color, H, W, "color")
    if col.shape[0] != 3:
        raise RuntimeError("hsr_utils.densify: color must be [3,H,W]")
    K = intrinsics.detach().float().cpu()
    c2w = torch.inverse(w2c.detach().float()).to(dev).contiguous()          # scripts/hierslam.py:167
    N = H * W
    o = dict(dtype=torch.float32, device=dev)
    means, rgb = torch.empty((N, 3), **o), torch.empty((N, 3), **o)
    ls, msd = torch.empty((N,), **o), torch.empty((N,), **o)
    mask = torch.empty((N,), dtype=torch.uint8, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    sc = torch.empty(int(_lib.hsr_densify_scratch_bytes(H, W)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.hsr_densify_frame(H, W, sil.data_ptr(), rd.data_ptr(), gt.data_ptr(), col.data_ptr(), float(K[0, 0]), float(K[1, 1]),
                                    float(K[0, 2]), float(K[1, 2]), c2w.data_ptr(), float(sil_thres), float(depth_factor), N,
                                    count.data_ptr(), means.data_ptr(), rgb.data_ptr(), ls.data_ptr(), msd.data_ptr(), mask.data_ptr(),
                                    None, sc.data_ptr(), sc.numel(), torch.cuda.current_stream(dev).cuda_stream)
    if rc < 0:
        _glue._fail(rc, "hsr_densify_frame")
    M = int(count.item())
    return torch.cat((means[:M], rgb[:M]), dim=1), msd[:M], mask.bool(), ls[:M, None]


def initialize_new_params_semantic(new_pt_cld, mean3_sq_dist, num_labels, log_scales=None):
    """scripts/hierslam.py:1137-1167 (flag_init = 2: semantic logits ~ U(0,1))."""
    num_pts = new_pt_cld.shape[0]
    dev = new_pt_cld.device
    unnorm_rots = torch.zeros((num_pts, 4), dtype=torch.float32, device=dev)
    unnorm_rots[:, 0] = 1.0
    params = {
        'means3D': new_pt_cld[:, :3],
        'rgb_colors': new_pt_cld[:, 3:6],
        'unnorm_rotations': unnorm_rots,
        'logit_opacities': torch.zeros((num_pts, 1), dtype=torch.float32, device=dev),
        'log_scales': log_scales if log_scales is not None else torch.log(torch.sqrt(mean3_sq_dist))[..., None],
        'semantic': torch.rand((num_pts, num_labels), device=dev),
    }
    return {k: torch.nn.Parameter(v.float().contiguous().requires_grad_(True)) for k, v in params.items()}


def initialize_new_params(new_pt_cld, mean3_sq_dist, gaussian_distribution, log_scales=None):
    """scripts/hierslam.py:1110-1135 (the map without semantic logits)."""
    num_pts = new_pt_cld.shape[0]
    dev = new_pt_cld.device
    unnorm_rots = torch.zeros((num_pts, 4), dtype=torch.float32, device=dev)
    unnorm_rots[:, 0] = 1.0
    ls = log_scales if log_scales is not None else torch.log(torch.sqrt(mean3_sq_dist))[..., None]
    if gaussian_distribution == "isotropic":
        ls = torch.tile(ls.reshape(-1, 1), (1, 1))
    elif gaussian_distribution == "anisotropic":
        ls = torch.tile(ls.reshape(-1, 1), (1, 3))
    else:
        raise ValueError(f"Unknown gaussian_distribution {gaussian_distribution}")
    params = {
        'means3D': new_pt_cld[:, :3],
        'rgb_colors': new_pt_cld[:, 3:6],
        'unnorm_rotations': unnorm_rots,
        'logit_opacities': torch.zeros((num_pts, 1), dtype=torch.float32, device=dev),
        'log_scales': ls,
    }
    return {k: torch.nn.Parameter(v.float().contiguous().requires_grad_(True)) for k, v in params.items()}


def _frame_w2c(params, time_idx):
    """the frame's world-to-camera from the pose parameters (scripts/hierslam.py:1283-1287; build_rotation:
    utils/slam_external.py:25-42)"""
    cam_rot = torch.nn.functional.normalize(params['cam_unnorm_rots'][..., time_idx].detach())
    curr_w2c = torch.eye(4, device=cam_rot.device)
    r, x, y, z = (cam_rot / cam_rot.norm(dim=1, keepdim=True))[0]
    curr_w2c[:3, :3] = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)]),
                                    torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)]),
                                    torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)])])
    curr_w2c[:3, 3] = params['cam_trans'][0, :, time_idx].detach()
    return curr_w2c


def _grow_map(params, variables, new_params, time_idx, n_new):
    """scripts/hierslam.py:1253-1260 / :1297-1304 / :1343-1350: concatenate, reset the densification statistics, stamp the
    new points with the frame index.  (With an optimizer attached the reference rebuilds it afterwards, :1655-1666; the
    optimizer-preserving form is hsr_utils.slam_external.cat_params_to_optimizer.)"""
    for k, v in new_params.items():
        params[k] = torch.nn.Parameter(torch.cat((params[k], v), dim=0).requires_grad_(True))
    num_pts = params['means3D'].shape[0]
    dev = params['means3D'].device
    variables['means2D_gradient_accum'] = torch.zeros(num_pts, device=dev).float()
    variables['denom'] = torch.zeros(num_pts, device=dev).float()
    variables['max_2D_radius'] = torch.zeros(num_pts, device=dev).float()
    new_timestep = time_idx * torch.ones(n_new, device=dev).float()
    variables['timestep'] = torch.cat((variables['timestep'], new_timestep), dim=0)
    return params, variables


def add_new_gaussians_semantic(params, variables, curr_data, sil_thres, time_idx, mean_sq_dist_method, num_semantic):
    """scripts/hierslam.py:1264-1305 (silhouette from the depth+silhouette render; NOT what the reference's loop calls — it
    calls the two functions below).  curr_data: 'cam', 'w2c', 'depth' [1,H,W], 'im' [3,H,W], 'intrinsics' [3,3]."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from . import slam_helpers as SH
    if mean_sq_dist_method != "projective":
        raise ValueError(f"Unknown mean_sq_dist_method {mean_sq_dist_method}")           # :178-179
    tg = SH.transform_to_frame(params, time_idx, gaussians_grad=False, camera_grad=False)
    rv = SH.transformed_params2depthplussilhouette(params, curr_data['w2c'], tg)
    with torch.no_grad():
        depth_sil = Renderer(raster_settings=curr_data['cam'])(**rv)[0]
        new_pt_cld, mean3_sq_dist, mask, log_scales = non_presence_points(depth_sil[1], depth_sil[0], curr_data['depth'][0], curr_data['im'],
                                                                          curr_data['intrinsics'], _frame_w2c(params, time_idx), sil_thres)
    if new_pt_cld.shape[0] > 0:
        new_params = initialize_new_params_semantic(new_pt_cld, mean3_sq_dist, num_semantic, log_scales)
        params, variables = _grow_map(params, variables, new_params, time_idx, new_pt_cld.shape[0])
    return params, variables


def add_new_gaussians_semantic_newrender(params, variables, curr_data, sil_thres, time_idx, mean_sq_dist_method, num_semantic,
                                         flag_use_render=1):
    """scripts/hierslam.py:1307-1352 — what the mapping loop calls for semantic maps (:1949): ONE no-grad render with the
    SEMANTIC rasterizer; silhouette = its final opacity (:1317), depth = its alpha-blended depth (:1322)."""
    from diff_gaussian_rasterization import GaussianRasterizer_semantic as Renderer_semantic
    from . import slam_helpers as SH
    if mean_sq_dist_method != "projective":
        raise ValueError(f"Unknown mean_sq_dist_method {mean_sq_dist_method}")
    if flag_use_render != 1:
        raise ValueError("flag_use_render must be 1 (the reference defines no other branch, scripts/hierslam.py:1313-1315)")
    tg = SH.transform_to_frame(params, time_idx, gaussians_grad=False, camera_grad=False)
    rv = SH.transformed_params2rendervar_semantic(params, tg)
    with torch.no_grad():
        im, radius, im_semantic, rendered_depth, rendered_median_depth, rendered_final_opcity = \
            Renderer_semantic(raster_settings=curr_data['cam'])(**rv)
        new_pt_cld, mean3_sq_dist, mask, log_scales = non_presence_points(
            rendered_final_opcity.squeeze(0), rendered_depth.squeeze(0), curr_data['depth'][0], curr_data['im'],
            curr_data['intrinsics'], _frame_w2c(params, time_idx), sil_thres)
    if new_pt_cld.shape[0] > 0:
        new_params = initialize_new_params_semantic(new_pt_cld, mean3_sq_dist, num_semantic, log_scales)
        params, variables = _grow_map(params, variables, new_params, time_idx, new_pt_cld.shape[0])
    return params, variables


def add_new_gaussians_newtest(params, variables, curr_data, sil_thres, time_idx, mean_sq_dist_method, gaussian_distribution,
                              flag_use_render=1):
    """scripts/hierslam.py:1214-1262 — what the mapping loop calls for maps without semantics (:1945): one no-grad render with
    the plain rasterizer (6 outputs); silhouette = final opacity (:1225), depth = alpha-blended depth (:1230)."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from . import slam_helpers as SH
    if mean_sq_dist_method != "projective":
        raise ValueError(f"Unknown mean_sq_dist_method {mean_sq_dist_method}")
    if flag_use_render != 1:
        raise ValueError("flag_use_render=2 unpacks 4 outputs from a 6-output renderer in the reference (:1221-1222) and cannot run")
    tg = SH.transform_to_frame(params, time_idx, gaussians_grad=False, camera_grad=False)
    rv = SH.transformed_params2rendervar(params, tg)
    with torch.no_grad():
        im, radius, rendered_depth, rendered_median_depth, rendered_final_opcity, rendered_mask = \
            Renderer(raster_settings=curr_data['cam'])(**rv)
        new_pt_cld, mean3_sq_dist, mask, log_scales = non_presence_points(
            rendered_final_opcity[0], rendered_depth[0], curr_data['depth'][0], curr_data['im'], curr_data['intrinsics'],
            _frame_w2c(params, time_idx), sil_thres)
    if new_pt_cld.shape[0] > 0:
        new_params = initialize_new_params(new_pt_cld, mean3_sq_dist, gaussian_distribution, log_scales)
        params, variables = _grow_map(params, variables, new_params, time_idx, new_pt_cld.shape[0])
    return params, variables
