"""Silhouette densification of a mapped frame, with the reference's function names.

    non_presence_points(...)                 mask (scripts/hierslam.py:1271-1278, :1289-1290) + get_pointcloud(..., mask=...,
                                             compute_mean_sq_dist=True) (:144-194) in one device pipeline (include/hsr_densify.h)
    initialize_new_params_semantic(...)      scripts/hierslam.py:1137-1167
    add_new_gaussians_semantic(...)          scripts/hierslam.py:1264-1305 (renders depth + silhouette with the fused input
                                             preparation and the rasterizer of this repo, then the two functions above)

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


def add_new_gaussians_semantic(params, variables, curr_data, sil_thres, time_idx, mean_sq_dist_method, num_semantic):
    """scripts/hierslam.py:1264-1305.  curr_data: 'cam', 'w2c', 'depth' [1,H,W], 'im' [3,H,W], 'intrinsics' [3,3]."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from . import slam_helpers as SH
    if mean_sq_dist_method != "projective":
        raise ValueError(f"Unknown mean_sq_dist_method {mean_sq_dist_method}")           # :178-179
    tg = SH.transform_to_frame(params, time_idx, gaussians_grad=False, camera_grad=False)
    rv = SH.transformed_params2depthplussilhouette(params, curr_data['w2c'], tg)
    with torch.no_grad():
        depth_sil = Renderer(raster_settings=curr_data['cam'])(**rv)[0]
        cam_rot = torch.nn.functional.normalize(params['cam_unnorm_rots'][..., time_idx].detach())
        curr_w2c = torch.eye(4, device=cam_rot.device)
        r, x, y, z = (cam_rot / cam_rot.norm(dim=1, keepdim=True))[0]
        curr_w2c[:3, :3] = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)]),
                                        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)]),
                                        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)])])
        curr_w2c[:3, 3] = params['cam_trans'][0, :, time_idx].detach()
        new_pt_cld, mean3_sq_dist, mask, log_scales = non_presence_points(depth_sil[1], depth_sil[0], curr_data['depth'][0], curr_data['im'],
                                                                          curr_data['intrinsics'], curr_w2c, sil_thres)
    if new_pt_cld.shape[0] > 0:
        new_params = initialize_new_params_semantic(new_pt_cld, mean3_sq_dist, num_semantic, log_scales)
        for k, v in new_params.items():
            params[k] = torch.nn.Parameter(torch.cat((params[k], v), dim=0).requires_grad_(True))
        num_pts = params['means3D'].shape[0]
        dev = params['means3D'].device
        variables['means2D_gradient_accum'] = torch.zeros(num_pts, device=dev).float()
        variables['denom'] = torch.zeros(num_pts, device=dev).float()
        variables['max_2D_radius'] = torch.zeros(num_pts, device=dev).float()
        new_timestep = time_idx * torch.ones(new_pt_cld.shape[0], device=dev).float()
        variables['timestep'] = torch.cat((variables['timestep'], new_timestep), dim=0)
    return params, variables
