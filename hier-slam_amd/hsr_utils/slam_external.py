"""Optimizer surgery of the map, with the reference's function names and semantics
(utils/slam_external.py):

    remove_points(to_remove, params, variables, optimizer)            :139-165
    prune_gaussians(params, variables, optimizer, iter, prune_dict)   :167-188
    cat_params_to_optimizer(new_params, params, optimizer)            :121-137
    update_params_and_optimizer(new_params, params, optimizer)        :107-119
    inverse_sigmoid, accumulate_mean2d_gradient                       :163-164, :100-104
    densify(params, variables, optimizer, iter, densify_dict)         :191-242   (gradient-driven clone / split / prune; off in the
                                                                                   reference's configs, kept for callers that enable it)

The reference prunes with ~22 boolean-mask gathers (six parameters, two Adam moments each, four bookkeeping vectors), each
with its own nonzero() and host sync, and concatenates tensor by tensor.  Here every per-Gaussian tensor is a row table of
ONE order-preserving device compaction (include/hsr_densify.h: hsr_prune_mask + hsr_compact_append_rows): one mask kernel,
one scan, one copy kernel, one read-back of the new row count.  Results are bit-identical to the reference's (pinned by
tests/golden/densify_prune_concat.npz, which holds the reference's own outputs).  There is no CPU path.
"""
import ctypes as C

import torch

from diff_gaussian_rasterization import _C as _glue

_lib = _glue._lib
_vp, _ci, _cf, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
MAX_TABLES = 40   # HSR_MAX_ROW_TABLES


class _RowTable(C.Structure):
    _fields_ = [("src", _vp), ("append", _vp), ("dst", _vp), ("cols", _ci)]


_lib.hsr_compact_scratch_bytes.restype = _sz
_lib.hsr_compact_scratch_bytes.argtypes = [_ci]
_lib.hsr_prune_mask.restype = _ci
_lib.hsr_prune_mask.argtypes = [_ci, _ci, _vp, _vp, _cf, _cf, _vp, _vp, _vp, _sz, _vp]
_lib.hsr_compact_append_rows.restype = _ci
_lib.hsr_compact_append_rows.argtypes = [_ci, _vp, _ci, _ci, C.POINTER(_RowTable), _ci, _vp, _vp, _sz, _vp]

CAMERA_KEYS = ('cam_unnorm_rots', 'cam_trans')     # not per-Gaussian: never pruned or extended (:141)
VARIABLE_KEYS = ('means2D_gradient_accum', 'denom', 'max_2D_radius', 'timestep')   # :160-164


def _rows2d(t, what):
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError("hsr_utils.slam_external: %s must be a float32 tensor on a HIP device; there is no CPU path" % what)
    t = t.detach().contiguous()
    return t.reshape(t.shape[0], -1) if t.dim() != 1 else t.reshape(-1, 1)


def compact_append(tensors, keep=None, appended=None, n_append=0, scanned=None):
    """One fused launch over a list of per-Gaussian tensors [P, ...]: rows with keep != 0 in source order (all rows if keep
    is None), then n_append new rows per tensor (appended[i], or zeros where it is None).  Returns the list of new tensors
    (same trailing shapes) — one host read-back for the row count when a mask is given."""
    if not tensors:
        return []
    dev = tensors[0].device
    P = int(tensors[0].shape[0])
    out, tabs, hold = [], (_RowTable * len(tensors))(), []
    if len(tensors) > MAX_TABLES:
        raise RuntimeError("at most %d tensors per compaction" % MAX_TABLES)
    for i, t in enumerate(tensors):
        if int(t.shape[0]) != P:
            raise RuntimeError("compact_append: tensor %d has %d rows, expected %d" % (i, t.shape[0], P))
        src = _rows2d(t, "tensor %d" % i)
        cols = int(src.shape[1])
        app = None
        if appended is not None and appended[i] is not None and n_append:
            app = _rows2d(appended[i], "appended tensor %d" % i)
            if tuple(app.shape) != (n_append, cols):
                raise RuntimeError("compact_append: appended tensor %d is %s, expected (%d, %d)" % (i, tuple(app.shape), n_append, cols))
        dst = torch.empty((P + n_append, cols), dtype=torch.float32, device=dev)
        hold += [src, app]
        out.append(dst)
        tabs[i] = _RowTable(src.data_ptr() if P else None, app.data_ptr() if app is not None else None, dst.data_ptr(), cols)
    if scanned is not None:
        scratch, rows_dev = scanned
    else:
        scratch = torch.empty(int(_lib.hsr_compact_scratch_bytes(P)), dtype=torch.uint8, device=dev)
        rows_dev = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.hsr_compact_append_rows(P, keep.data_ptr() if keep is not None else None, 1 if scanned is not None else 0,
                                          len(tensors), tabs, int(n_append), rows_dev.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                          torch.cuda.current_stream(dev).cuda_stream)
    if rc < 0:
        _glue._fail(rc, "hsr_compact_append_rows")
    rows = P + n_append if keep is None else int(rows_dev.item())
    return [o[:rows].reshape((rows,) + tuple(t.shape[1:])) for o, t in zip(out, tensors)]


def _gaussian_tables(params, variables, optimizer):
    """(names, tensors, kinds): every per-Gaussian tensor the surgery touches, in a fixed order"""
    names, tensors = [], []
    for k in params.keys():
        if k in CAMERA_KEYS:
            continue
        group = [g for g in optimizer.param_groups if g['name'] == k][0]
        p = group['params'][0]
        names.append(("param", k)); tensors.append(p.data)
        st = optimizer.state.get(p, None)
        if st is not None and "exp_avg" in st:
            names.append(("exp_avg", k)); tensors.append(st["exp_avg"])
            names.append(("exp_avg_sq", k)); tensors.append(st["exp_avg_sq"])
    for k in VARIABLE_KEYS:
        if variables is not None and k in variables and torch.is_tensor(variables[k]):
            names.append(("var", k)); tensors.append(variables[k])
    return names, tensors


def _install(names, new_tensors, params, variables, optimizer):
    """puts the rebuilt tensors back the way the reference does: a fresh nn.Parameter per group, its Adam state re-keyed
    (:148-153), variables replaced (:160-164)"""
    by = {n: t for n, t in zip(names, new_tensors)}
    for k in list(params.keys()):
        if ("param", k) not in by:
            continue
        group = [g for g in optimizer.param_groups if g['name'] == k][0]
        old = group['params'][0]
        st = optimizer.state.get(old, None)
        new_p = torch.nn.Parameter(by[("param", k)].requires_grad_(True))
        if st is not None:
            if ("exp_avg", k) in by:
                st["exp_avg"] = by[("exp_avg", k)]
                st["exp_avg_sq"] = by[("exp_avg_sq", k)]
            del optimizer.state[old]
            optimizer.state[new_p] = st
        group['params'][0] = new_p
        params[k] = new_p
    if variables is not None:
        for k in VARIABLE_KEYS:
            if ("var", k) in by:
                variables[k] = by[("var", k)]


def remove_points(to_remove, params, variables, optimizer, _scanned=None):
    """utils/slam_external.py:139-165 — `to_remove` is a bool [P] tensor on the device."""
    keep = (~to_remove).to(torch.uint8).contiguous() if _scanned is None else to_remove
    names, tensors = _gaussian_tables(params, variables, optimizer)
    new = compact_append(tensors, keep=keep, scanned=_scanned)
    _install(names, new, params, variables, optimizer)
    return params, variables


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def update_params_and_optimizer(new_params, params, optimizer):
    """Whole-tensor replacement of parameters (the opacity reset) with their Adam moments restarted from zero — what
    utils/slam_external.py:107-119 does per key — expressed on the same (name, tensor) table and installer as the prune /
    append compaction above: the new value becomes the parameter, zeros become its two moments, the state keeps its step."""
    names, tensors = [], []
    for key, value in new_params.items():
        names.append(("param", key)); tensors.append(value)
        names.append(("exp_avg", key)); tensors.append(torch.zeros_like(value))
        names.append(("exp_avg_sq", key)); tensors.append(torch.zeros_like(value))
    _install(names, tensors, params, None, optimizer)
    return params


def prune_mask(params, variables, removal_opacity_threshold, remove_big):
    """(keep uint8 [P], scanned state) of :175-180 from one kernel + scan; `remove_big` adds the 0.1 * scene_radius test."""
    lo = params['logit_opacities']
    ls = params['log_scales']
    dev = lo.device
    P = int(lo.shape[0])
    big = 0.0
    if remove_big:
        big = float(0.1 * variables['scene_radius'])        # the reference's own expression (:179), evaluated by torch
    keep = torch.empty(P, dtype=torch.uint8, device=dev)
    scratch = torch.empty(int(_lib.hsr_compact_scratch_bytes(P)), dtype=torch.uint8, device=dev)
    kept = torch.empty(1, dtype=torch.int32, device=dev)
    lo2, ls2 = _rows2d(lo, "logit_opacities"), _rows2d(ls, "log_scales")
    with torch.cuda.device(dev):
        rc = _lib.hsr_prune_mask(P, int(ls2.shape[1]), lo2.data_ptr() if P else None, ls2.data_ptr() if P else None,
                                 float(removal_opacity_threshold), float(big), keep.data_ptr() if P else None, kept.data_ptr(),
                                 scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream(dev).cuda_stream)
    if rc < 0:
        _glue._fail(rc, "hsr_prune_mask")
    return keep, (scratch, kept)


def _prune_schedule(it, cfg):
    """what utils/slam_external.py:167-188 does at mapping iteration `it` under the schedule `cfg` (the reference's keys):
    (opacity threshold to prune at | None, also remove big Gaussians, reset opacities afterwards)"""
    if it > cfg['stop_after']:
        return None, False, False
    threshold = None
    if it >= cfg['start_after'] and it % cfg['prune_every'] == 0:
        threshold = cfg['final_removal_opacity_threshold'] if it == cfg['stop_after'] else cfg['removal_opacity_threshold']
    reset = bool(cfg['reset_opacities']) and it > 0 and it % cfg['reset_opacities_every'] == 0
    return threshold, it >= cfg['remove_big_after'], reset


def prune_gaussians(params, variables, optimizer, iter, prune_dict):
    """utils/slam_external.py:167-188, same schedule keys (start_after, remove_big_after, stop_after, prune_every,
    removal_opacity_threshold, final_removal_opacity_threshold, reset_opacities, reset_opacities_every): one fused mask + scan
    (hsr_prune_mask) and one row compaction over every parameter, Adam moment and bookkeeping vector (hsr_compact_append_rows)
    instead of ~22 boolean-mask gathers."""
    threshold, remove_big, reset = _prune_schedule(iter, prune_dict)
    if threshold is not None:
        keep, scanned = prune_mask(params, variables, threshold, remove_big)
        params, variables = remove_points(keep, params, variables, optimizer, _scanned=scanned)
    if reset:   # every opacity back to 0.01 (:186), moments restarted
        params = update_params_and_optimizer({'logit_opacities': inverse_sigmoid(torch.full_like(params['logit_opacities'], 0.01))},
                                             params, optimizer)
    return params, variables


def cat_params_to_optimizer(new_params, params, optimizer):
    """:121-137 — appends the new Gaussians to every parameter and zeros to its Adam moments, one fused copy."""
    keys = [k for k in new_params.keys()]
    names, tensors, app = [], [], []
    n_new = int(next(iter(new_params.values())).shape[0]) if keys else 0
    for k in keys:
        group = [g for g in optimizer.param_groups if g['name'] == k][0]
        p = group['params'][0]
        v = new_params[k].detach()
        names.append(("param", k)); tensors.append(p.data); app.append(v)
        st = optimizer.state.get(p, None)
        if st is not None and "exp_avg" in st:
            names.append(("exp_avg", k)); tensors.append(st["exp_avg"]); app.append(None)
            names.append(("exp_avg_sq", k)); tensors.append(st["exp_avg_sq"]); app.append(None)
    new = compact_append(tensors, keep=None, appended=app, n_append=n_new)
    _install(names, new, params, None, optimizer)
    return params


def accumulate_mean2d_gradient(variables):
    """:100-104"""
    variables['means2D_gradient_accum'][variables['seen']] += torch.norm(variables['means2D'].grad[variables['seen'], :2], dim=-1)
    variables['denom'][variables['seen']] += 1
    return variables


def _rotate_by_quaternion(q, v):
    """R(q / |q|) v for rows of quaternions (r, x, y, z) and vectors: v + 2 r (u x v) + 2 u x (u x v), u = (x, y, z)"""
    q = q / q.norm(dim=1, keepdim=True)
    r, u = q[:, :1], q[:, 1:]
    c = torch.cross(u, v, dim=1)
    return v + 2.0 * (r * c + torch.cross(u, c, dim=1))


def densify(params, variables, optimizer, iter, densify_dict):
    """utils/slam_external.py:191-242 — the 3DGS-style densification the reference keeps behind `use_gaussian_splatting_densification`
    (False in its configs): while iter <= stop_after the image-plane gradient norm of every seen Gaussian is accumulated; on a
    densification iteration (iter >= start_after, every densify_every) Gaussians whose mean accumulated gradient reaches grad_thresh are
    CLONED when small (largest scale <= 1 % of the scene radius) or SPLIT into num_to_split_into samples of themselves when large (means
    drawn from the Gaussian itself, scales divided by 0.8 n, the original removed), then everything below the opacity threshold — and,
    from remove_big_after on, above 10 % of the scene radius — is pruned and the accumulators restart; opacities are reset on their own
    schedule.  Same keys, same order of the resulting rows (kept originals, clones, splits) and the same single `torch.normal` draw as
    the reference, so a seeded run consumes the generator identically.

    Where the reference runs three concatenations and two removals tensor by tensor (each with its own mask gathers and host syncs),
    this decides every row's fate first and runs ONE fused compaction + append over all parameters, Adam moments and bookkeeping vectors
    (appended rows get zero moments, like `cat_params_to_optimizer`).  `variables['timestep']`, which the reference's version cannot
    carry through its concatenations, is inherited from the source Gaussian."""
    if iter > densify_dict['stop_after']:
        return params, variables
    variables = accumulate_mean2d_gradient(variables)
    if iter >= densify_dict['start_after'] and iter % densify_dict['densify_every'] == 0:
        with torch.no_grad():
            n = int(densify_dict['num_to_split_into'])
            radius = variables['scene_radius']
            mean_grad = variables['means2D_gradient_accum'] / variables['denom']
            mean_grad = torch.where(torch.isnan(mean_grad), torch.zeros_like(mean_grad), mean_grad)
            extent = torch.exp(params['log_scales']).max(dim=1).values
            hot = mean_grad >= densify_dict['grad_thresh']
            small = extent <= 0.01 * radius
            cloned, split = hot & small, hot & ~small
            opacity_floor = (densify_dict['final_removal_opacity_threshold'] if iter == densify_dict['stop_after']
                             else densify_dict['removal_opacity_threshold'])
            cull_big = iter >= densify_dict['remove_big_after']

            def doomed(logit_opacities, log_scales):   # the closing prune, applied to whatever set of rows
                out = (torch.sigmoid(logit_opacities) < opacity_floor).reshape(-1)
                if cull_big:
                    out = out | (torch.exp(log_scales).max(dim=1).values > 0.1 * radius)
                return out

            gaussian_keys = [k for k in params.keys() if k not in CAMERA_KEYS]
            fresh = {}
            for k in gaussian_keys:
                v = params[k].detach()
                fresh[k] = torch.cat((v[cloned], v[split].repeat(n, *([1] * (v.dim() - 1)))), dim=0)
            n_clone, n_split = int(cloned.sum()), int(split.sum())
            if n_split:
                sigma = torch.exp(params['log_scales'].detach()[split])
                sigma = (sigma.expand(-1, 3) if sigma.shape[1] == 1 else sigma).repeat(n, 1)
                offsets = torch.normal(mean=torch.zeros_like(sigma), std=sigma)
                turn = params['unnorm_rotations'].detach()[split].repeat(n, 1)
                fresh['means3D'][n_clone:] += _rotate_by_quaternion(turn, offsets)
                fresh['log_scales'][n_clone:] = torch.log(torch.exp(fresh['log_scales'][n_clone:]) / (0.8 * n))
            stays = ~doomed(fresh['logit_opacities'], fresh['log_scales'])
            fresh = {k: v[stays].contiguous() for k, v in fresh.items()}
            n_new = int(stays.sum())
            keep = (~split & ~doomed(params['logit_opacities'].detach(), params['log_scales'].detach())).to(torch.uint8).contiguous()

            names, tensors = _gaussian_tables(params, variables, optimizer)
            source_time = None
            if 'timestep' in variables and torch.is_tensor(variables['timestep']):
                t = variables['timestep']
                source_time = torch.cat((t[cloned], t[split].repeat(n)))[stays].to(torch.float32).contiguous()
            appended = []
            for kind, k in names:
                if kind == "param":
                    appended.append(fresh[k])
                elif kind == "var" and k == 'timestep':
                    appended.append(source_time)
                else:
                    appended.append(None)      # Adam moments and accumulators of new rows start at zero
            new = compact_append(tensors, keep=keep, appended=appended, n_append=n_new)
            _install(names, new, params, variables, optimizer)
            rows = int(params['means3D'].shape[0])
            dev = params['means3D'].device
            for k in ('means2D_gradient_accum', 'denom', 'max_2D_radius'):
                variables[k] = torch.zeros(rows, device=dev)
    if iter > 0 and iter % densify_dict['reset_opacities_every'] == 0 and densify_dict['reset_opacities']:
        params = update_params_and_optimizer({'logit_opacities': inverse_sigmoid(torch.full_like(params['logit_opacities'], 0.01))},
                                             params, optimizer)
    return params, variables
