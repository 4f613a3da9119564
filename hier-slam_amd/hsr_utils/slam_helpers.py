"""Rasterizer-input preparation with the reference's function names, fused into one HIP launch per direction.

Mirror of the four caller-side helpers scripts/hierslam.py runs right before every render:

    transform_to_frame(params, time_idx, gaussians_grad, camera_grad)          utils/slam_helpers.py:278-330
    transformed_params2rendervar(params, transformed_gaussians)                utils/slam_helpers.py:124-139
    transformed_params2rendervar_semantic(params, transformed_gaussians)       utils/slam_helpers.py:195-219
    transformed_params2depthplussilhouette(params, w2c, transformed_gaussians) utils/slam_helpers.py:260-275

Same names, arguments, dictionary keys, shapes and gradient flow (`gaussians_grad` / `camera_grad` detach exactly what
the reference detaches).  In the reference each is a chain of small torch kernels (about 12 launches forward, 25
backward, per iteration); here `transform_to_frame` returns a lazy dictionary and the `…2rendervar…` call that follows
runs ONE kernel (`hsr_frame_prep_forward`, include/hsr_frame_prep.h) producing every tensor of both dictionaries; the
backward is one per-Gaussian kernel plus a one-block finish of the pose-gradient reduction.  Touching
`transformed_gaussians['means3D']` before any rendervar call simply runs the same kernel earlier.

There is no CPU path: tensors must live on a HIP device and libhsr_rast.so must be present (ImportError otherwise).
Deviation kept on purpose: for anisotropic Gaussians (log_scales [P,3]) the reference's semantic variant tiles the
scales to [P,9] (slam_helpers.py:215) and the rasterizer then reads out of bounds; here scales stay [P,3].
"""
import ctypes as C

import torch

from diff_gaussian_rasterization import _C as _glue

ROT_PARAMS, ROT_TRANSFORMED = 0, 1

_lib = _glue._lib
_vp, _ci, _sz = C.c_void_p, C.c_int, C.c_size_t
_lib.hsr_frame_prep_scratch_bytes.restype = _sz
_lib.hsr_frame_prep_scratch_bytes.argtypes = [_ci]
_lib.hsr_frame_prep_forward.restype = _ci
_lib.hsr_frame_prep_forward.argtypes = [_ci, _ci, _ci, _ci] + [_vp] * 6 + [_ci, _ci] + [_vp] * 8
_lib.hsr_frame_prep_backward.restype = _ci
_lib.hsr_frame_prep_backward.argtypes = [_ci, _ci, _ci, _ci] + [_vp] * 6 + [_ci, _ci] + [_vp] * 14 + [_sz, _vp]
_lib.hsr_frame_prep_backward_params.restype = _ci
_lib.hsr_frame_prep_backward_params.argtypes = _lib.hsr_frame_prep_backward.argtypes


def _dev_f32(t, what):
    if not t.is_cuda:
        raise RuntimeError("hsr_utils.slam_helpers: %s must live on a HIP device (got %s); there is no CPU path" % (what, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("hsr_utils.slam_helpers: %s must be float32 (got %s)" % (what, t.dtype))
    return t.contiguous()


def _p(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


class _FramePrep(torch.autograd.Function):
    """(means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rots, cam_trans) ->
    (means3D_cam, transformed_unnorm_rot, rotations, opacities, scales, depth_sil | empty)."""

    @staticmethod
    def forward(ctx, means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rots, cam_trans, w2c, time_idx,
                rot_source, gaussians_grad, camera_grad):
        means3D = _dev_f32(means3D, "means3D")
        dev = means3D.device
        unnorm_rotations = _dev_f32(unnorm_rotations, "unnorm_rotations")
        logit_opacities = _dev_f32(logit_opacities, "logit_opacities")
        log_scales = _dev_f32(log_scales, "log_scales")
        cam_unnorm_rots = _dev_f32(cam_unnorm_rots, "cam_unnorm_rots")
        cam_trans = _dev_f32(cam_trans, "cam_trans")
        P = int(means3D.shape[0])
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        if tuple(unnorm_rotations.shape) != (P, 4) or logit_opacities.numel() != P or log_scales.dim() != 2 or log_scales.shape[0] != P:
            raise RuntimeError("unnorm_rotations [P,4], logit_opacities [P,1] and log_scales [P,1|3] must match means3D")
        S = int(log_scales.shape[1])
        if cam_unnorm_rots.dim() != 3 or cam_unnorm_rots.shape[:2] != (1, 4) or cam_trans.shape[:2] != (1, 3) \
                or cam_trans.shape[2] != cam_unnorm_rots.shape[2]:
            raise RuntimeError("cam_unnorm_rots must be [1,4,num_frames] and cam_trans [1,3,num_frames]")
        frames = int(cam_unnorm_rots.shape[2])
        time_idx = int(time_idx)
        if time_idx < 0:
            time_idx += frames
        transform_rots = S != 1                                         # slam_helpers.py:302-306
        if w2c is not None:
            w2c = _dev_f32(w2c, "w2c")
            if tuple(w2c.shape) != (4, 4):
                raise RuntimeError("w2c must be [4,4]")
        o = dict(dtype=torch.float32, device=dev)
        out_means = torch.empty((P, 3), **o)
        out_tr = torch.empty((P, 4), **o)   # isotropic: a copy of unnorm_rotations (the reference hands back the same tensor)
        out_rot = torch.empty((P, 4), **o)
        out_op = torch.empty_like(logit_opacities)
        out_sc = torch.empty((P, 3), **o)
        out_sil = torch.empty((P, 3), **o) if w2c is not None else torch.empty(0, **o)
        with torch.cuda.device(dev):
            rc = _lib.hsr_frame_prep_forward(P, S, int(transform_rots), int(rot_source), _p(means3D), _p(unnorm_rotations),
                                             _p(logit_opacities), _p(log_scales), cam_unnorm_rots.data_ptr(), cam_trans.data_ptr(),
                                             frames, time_idx, None if w2c is None else w2c.data_ptr(), _p(out_means),
                                             _p(out_tr), _p(out_rot), _p(out_op), _p(out_sc),
                                             _p(out_sil) if w2c is not None else None, torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            _glue._fail(rc, "hsr_frame_prep_forward")
        ctx.save_for_backward(means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rots, cam_trans,
                              w2c if w2c is not None else torch.empty(0, **o))
        ctx.meta = (P, S, transform_rots, int(rot_source), frames, time_idx, bool(gaussians_grad), bool(camera_grad), w2c is not None)
        ctx.set_materialize_grads(False)    # unused outputs arrive as None instead of zero tensors
        return out_means, out_tr, out_rot, out_op, out_sc, out_sil

    @staticmethod
    def backward(ctx, g_means, g_tr, g_rot, g_op, g_sc, g_sil):
        means3D, unnorm_rotations, logit_opacities, log_scales, cam_unnorm_rots, cam_trans, w2c = ctx.saved_tensors
        P, S, transform_rots, rot_source, frames, time_idx, gaussians_grad, camera_grad, has_sil = ctx.meta
        dev = means3D.device
        o = dict(dtype=torch.float32, device=dev)

        def up(g):
            return None if g is None else g.contiguous().float()
        g_means, g_tr, g_rot, g_op, g_sc = up(g_means), up(g_tr), up(g_rot), up(g_op), up(g_sc)
        g_sil = up(g_sil) if has_sil else None
        # gradient sink (diff_gaussian_rasterization/_C.py set_gradient_sink): the parameter gradients may be written straight into
        # views of a communication bucket (hsr_utils/parallel.py GradientExchange)
        sink = (lambda name, shape: _glue._from_sink("params." + name, shape, dev)) if gaussians_grad else (lambda name, shape: None)

        def alloc(name, shape):
            t = sink(name, shape)
            return t if t is not None else torch.empty(shape, **o)
        d_means = alloc("means3D", (P, 3))
        d_unnorm = alloc("unnorm_rotations", (P, 4))
        d_logit = alloc("logit_opacities", tuple(logit_opacities.shape))
        d_ls = alloc("log_scales", tuple(log_scales.shape))
        # pose gradients shaped like the parameters: the finish kernel zeroes the other frames' columns itself
        d_rots = torch.empty_like(cam_unnorm_rots) if camera_grad else None
        d_trans = torch.empty_like(cam_trans) if camera_grad else None
        scratch = torch.empty(int(_lib.hsr_frame_prep_scratch_bytes(P)), dtype=torch.uint8, device=dev)

        def run(g_tr_arg, d_unnorm_out, cam_out):
            with torch.cuda.device(dev):
                rc = _lib.hsr_frame_prep_backward_params(
                    P, S, int(transform_rots), rot_source, _p(means3D), _p(unnorm_rotations), _p(logit_opacities), _p(log_scales),
                    cam_unnorm_rots.data_ptr(), cam_trans.data_ptr(), frames, time_idx, w2c.data_ptr() if has_sil else None,
                    _p(g_means), _p(g_tr_arg), _p(g_rot), _p(g_op), _p(g_sc), _p(g_sil), _p(d_means), _p(d_unnorm_out), _p(d_logit),
                    _p(d_ls), _p(d_rots) if cam_out else None, _p(d_trans) if cam_out else None, scratch.data_ptr(), scratch.numel(),
                    torch.cuda.current_stream(dev).cuda_stream)
            if rc < 0:
                _glue._fail(rc, "hsr_frame_prep_backward")

        run(g_tr, d_unnorm, camera_grad)
        if gaussians_grad:
            out_means, out_unnorm = d_means, d_unnorm
        else:
            # transform_to_frame detached the Gaussians (slam_helpers.py:312-314): only what the rendervar builders read
            # straight from params keeps its gradient — F.normalize(params['unnorm_rotations']) in the PARAMS variants.
            out_means = None
            if rot_source == ROT_PARAMS:
                if g_tr is not None:
                    d_unnorm = torch.empty((P, 4), **o)
                    run(None, d_unnorm, False)
                out_unnorm = d_unnorm
            else:
                out_unnorm = None
        return out_means, out_unnorm, d_logit, d_ls, d_rots, d_trans, None, None, None, None, None


class TransformedGaussians(dict):
    """What transform_to_frame returns: a dictionary with 'means3D' and 'unnorm_rotations' whose values are produced on
    first use, so that the rendervar builder that follows can choose the variant and get everything from one launch."""

    def __init__(self, params, time_idx, gaussians_grad, camera_grad):
        super().__init__()
        self._params, self._time_idx = params, time_idx
        self._gaussians_grad, self._camera_grad = bool(gaussians_grad), bool(camera_grad)
        self._bundle = None

    def _run(self, rot_source, w2c=None):
        p = self._params
        outs = _FramePrep.apply(p['means3D'], p['unnorm_rotations'], p['logit_opacities'], p['log_scales'], p['cam_unnorm_rots'],
                                p['cam_trans'], w2c, self._time_idx, rot_source, self._gaussians_grad, self._camera_grad)
        bundle = dict(zip(("means3D", "unnorm_rotations", "rotations", "opacities", "scales", "depth_sil"), outs))
        bundle["rot_source"], bundle["has_sil"] = rot_source, w2c is not None
        # which w2c the depth / silhouette plane was formed with: a later request with ANOTHER matrix (or the same tensor modified
        # in place) must not be served from this bundle
        bundle["w2c_key"] = None if w2c is None else (id(w2c), int(getattr(w2c, "_version", 0)))
        if self._bundle is None:
            self._bundle = bundle
            dict.__setitem__(self, 'means3D', bundle['means3D'])
            dict.__setitem__(self, 'unnorm_rotations', bundle['unnorm_rotations'])
        return bundle

    def bundle(self, rot_source, w2c=None):
        b = self._bundle
        if b is not None and b["rot_source"] == rot_source and (
                w2c is None or (b["has_sil"] and b["w2c_key"] == (id(w2c), int(getattr(w2c, "_version", 0))))):
            return b
        return self._run(rot_source, w2c)

    def _materialise(self):
        if self._bundle is None:
            self._run(ROT_TRANSFORMED)

    def __getitem__(self, k):
        self._materialise()
        return dict.__getitem__(self, k)

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __contains__(self, k):
        return k in ('means3D', 'unnorm_rotations')

    def __len__(self):
        return 2

    def __iter__(self):
        return iter(('means3D', 'unnorm_rotations'))


def transform_to_frame(params, time_idx, gaussians_grad, camera_grad):
    """World -> camera frame for frame `time_idx` (slam_helpers.py:278-330).  Returns a dictionary with 'means3D' and
    'unnorm_rotations' (lazy, see TransformedGaussians)."""
    return TransformedGaussians(params, time_idx, gaussians_grad, camera_grad)


def _bundle_of(params, transformed_gaussians, rot_source, w2c=None):
    if isinstance(transformed_gaussians, TransformedGaussians) and transformed_gaussians._params is params:
        return transformed_gaussians.bundle(rot_source, w2c)
    raise TypeError("transformed_gaussians must come from hsr_utils.slam_helpers.transform_to_frame(params, ...) "
                    "(the fused kernel produces both dictionaries in one launch)")


def _means2D(params):
    # the reference's gradient sink for densification statistics (slam_helpers.py:137)
    # (the reference adds 0 to turn the zeros into a non-leaf whose .grad it retains; a view does the same without a kernel: the
    # rasterizer never reads the values, it only returns this tensor's gradient)
    z = torch.zeros_like(params['means3D'], requires_grad=True)
    return z.view_as(z)


def transformed_params2rendervar(params, transformed_gaussians):
    b = _bundle_of(params, transformed_gaussians, ROT_TRANSFORMED)
    return {'means3D': b['means3D'], 'colors_precomp': params['rgb_colors'], 'rotations': b['rotations'],
            'opacities': b['opacities'], 'scales': b['scales'], 'means2D': _means2D(params)}


def transformed_params2rendervar_semantic(params, transformed_gaussians):
    b = _bundle_of(params, transformed_gaussians, ROT_PARAMS)
    return {'means3D': b['means3D'], 'colors_precomp': params['rgb_colors'], 'rotations': b['rotations'],
            'opacities': b['opacities'], 'scales': b['scales'], 'semantics_precomp': params['semantic'],
            'means2D': _means2D(params)}


def transformed_params2silhouette(params, transformed_gaussians):
    """slam_helpers.py:176-193: colour channel 0 = 1 (silhouette), rest as transformed_params2rendervar."""
    b = _bundle_of(params, transformed_gaussians, ROT_TRANSFORMED)
    sil_color = torch.zeros_like(params['rgb_colors'])
    sil_color[:, 0] = 1.0
    return {'means3D': b['means3D'], 'colors_precomp': sil_color, 'rotations': b['rotations'], 'opacities': b['opacities'],
            'scales': b['scales'], 'means2D': _means2D(params)}


def transformed_params2depthplussilhouette(params, w2c, transformed_gaussians):
    b = _bundle_of(params, transformed_gaussians, ROT_TRANSFORMED, w2c)
    return {'means3D': b['means3D'], 'colors_precomp': b['depth_sil'], 'rotations': b['rotations'], 'opacities': b['opacities'],
            'scales': b['scales'], 'means2D': _means2D(params)}
