"""Loss heads on the rendered maps with the reference's names, each fused into one streaming HIP pass that produces the
value and the gradient together (include/hsr_losses.h).

    l1_loss_v1(x, y)                                   utils/slam_helpers.py:5-6
    calc_ssim(img1, img2, window_size=11, size_average=True)      utils/slam_external.py:66-97
    masked_l1(pred, gt, mask, reduction)               torch.abs(gt - pred)[mask].sum() | .mean()   scripts/hierslam.py:921-937
    tree_cross_entropy(im_semantic, labels, num_semantic, weights=None)
                                                       sum over levels of CrossEntropyLoss(transfer_tree_rendered_labelmap(
                                                       im_semantic, i_level, dataset), labels[i_level])   scripts/hierslam.py:963-974
    cross_entropy_planar(logits, labels)               CrossEntropyLoss on [C,H,W] logits (flat classes, :947-954; leaf MLP, :976-983)

`mapping_image_loss(im, gt)` is the reference's mapping colour term 0.8*L1 + 0.2*(1 - SSIM) (scripts/hierslam.py:939).
Gradients flow to the FIRST argument only (the rendered map); the ground truth is data.  There is no CPU path.
"""
import ctypes as C

import torch

from diff_gaussian_rasterization import _C as _glue

_lib = _glue._lib
_vp, _ci, _sz = C.c_void_p, C.c_int, C.c_size_t
_lib.hsr_loss_scratch_bytes.restype = _sz
_lib.hsr_loss_scratch_bytes.argtypes = [_ci, _ci, _ci]
_lib.hsr_loss_l1.restype = _ci
_lib.hsr_loss_l1.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _vp, _vp, _sz, _vp]
_lib.hsr_loss_ssim.restype = _ci
_lib.hsr_loss_ssim.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _sz, _vp]
_lib.hsr_loss_l1_grad.restype = _ci
_lib.hsr_loss_l1_grad.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _vp, _vp]
_lib.hsr_loss_ssim_value.restype = _ci
_lib.hsr_loss_ssim_value.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _sz, _vp]
_lib.hsr_loss_ssim_grad.restype = _ci
_lib.hsr_loss_ssim_grad.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.hsr_loss_tree_ce.restype = _ci
_lib.hsr_loss_tree_ce.argtypes = [_ci, _ci, _ci, _ci, C.POINTER(_ci), C.POINTER(C.c_float), _vp, _vp, _ci, _vp, _vp, _vp, _sz, _vp]

_lib.hsr_loss_tree_ce_scratch_bytes.restype = _sz
_lib.hsr_loss_tree_ce_scratch_bytes.argtypes = [_ci, _ci]
_lib.hsr_loss_tree_ce_value.restype = _ci
_lib.hsr_loss_tree_ce_value.argtypes = [_ci, _ci, _ci, _ci, C.POINTER(_ci), _vp, _vp, _ci, _vp, _vp, _vp, _sz, _vp]
_lib.hsr_loss_tree_ce_grad.restype = _ci
_lib.hsr_loss_tree_ce_grad.argtypes = [_ci, _ci, _ci, _ci, C.POINTER(_ci), C.POINTER(C.c_float), _vp, _vp, _ci, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp]
_lib.hsr_loss_tracking_scratch_bytes.restype = _sz
_lib.hsr_loss_tracking_scratch_bytes.argtypes = [_ci, _ci]
_lib.hsr_loss_tracking_value.restype = _ci
_lib.hsr_loss_tracking_value.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, C.c_float, _ci, _ci, C.c_float, C.c_float, _vp, _vp, _sz, _vp]
_lib.hsr_loss_tracking_grad.restype = _ci
_lib.hsr_loss_tracking_grad.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, C.c_float, _ci, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp]
_lib.hsr_loss_leaf_mlp_ce.restype = _ci
_lib.hsr_loss_leaf_mlp_ce.argtypes = [_ci, _ci, _ci, _ci] + [_vp] * 4 + [_ci] + [_vp] * 5 + [_sz, _vp]

SUM, MEAN = 0, 1
LEAF_MAX_K, LEAF_MAX_C = 31, 128
_weight_cache = {}


def _chw(t, what):
    if not t.is_cuda:
        raise RuntimeError("hsr_utils.losses: %s must live on a HIP device (got %s); there is no CPU path" % (what, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("hsr_utils.losses: %s must be float32 (got %s)" % (what, t.dtype))
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3:
        raise RuntimeError("hsr_utils.losses: %s must be [C,H,W] or [H,W]" % what)
    return t.contiguous()


def _scratch(ch, H, W, dev):
    return torch.empty(int(_lib.hsr_loss_scratch_bytes(ch, H, W)), dtype=torch.uint8, device=dev)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mask, reduction):
        shape = pred.shape
        p, g = _chw(pred, "pred"), _chw(gt.detach(), "gt")
        if p.shape != g.shape:
            raise RuntimeError("hsr_utils.losses: pred %s and gt %s differ in shape" % (tuple(shape), tuple(gt.shape)))
        Cc, H, W = p.shape
        dev = p.device
        m = None
        if mask is not None:
            m = mask.reshape(-1, H, W)
            if m.shape[0] != 1:
                raise RuntimeError("hsr_utils.losses: mask must be [H,W] or [1,H,W] (it is tiled over the channels)")
            m = (m if m.dtype == torch.bool else m != 0).to(torch.uint8).contiguous()
        out = torch.empty(1, dtype=torch.float32, device=dev)
        # sums and the unmasked mean: value now, gradient in backward() times the incoming gradient (hsr_loss_l1_grad); the masked mean's
        # gradient needs the selection count of this pass and is stashed here, rescaled there
        ctx.two_pass = bool(pred.requires_grad) and (int(reduction) == SUM or m is None)
        grad = torch.empty_like(p) if (pred.requires_grad and not ctx.two_pass) else None
        sc = _scratch(Cc, H, W, dev)
        with torch.cuda.device(dev):
            rc = _lib.hsr_loss_l1(Cc, H, W, p.data_ptr(), g.data_ptr(), None if m is None else m.data_ptr(), int(reduction),
                                  out.data_ptr(), None if grad is None else grad.data_ptr(), sc.data_ptr(), sc.numel(), _stream(dev))
        if rc < 0:
            _glue._fail(rc, "hsr_loss_l1")
        ctx.grad = None if grad is None else grad.view(shape)
        if ctx.two_pass:
            ctx.save_for_backward(p, g, m if m is not None else torch.empty(0, device=dev))
            ctx.meta = (Cc, H, W, int(reduction), m is not None, tuple(shape))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if not ctx.two_pass or g is None:
            return (None if ctx.grad is None or g is None else ctx.grad * g), None, None, None
        p, gt, m = ctx.saved_tensors
        Cc, H, W, reduction, has_mask, shape = ctx.meta
        dev = p.device
        gg = g.to(device=dev, dtype=torch.float32).contiguous()
        grad = torch.empty_like(p)
        with torch.cuda.device(dev):
            rc = _lib.hsr_loss_l1_grad(Cc, H, W, p.data_ptr(), gt.data_ptr(), m.data_ptr() if has_mask else None, reduction, gg.data_ptr(),
                                       grad.data_ptr(), _stream(dev))
        if rc < 0:
            _glue._fail(rc, "hsr_loss_l1_grad")
        return grad.view(shape), None, None, None


class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        shape = img1.shape
        a, b = _chw(img1, "img1"), _chw(img2.detach(), "img2")
        if a.shape != b.shape:
            raise RuntimeError("hsr_utils.losses: img1 %s and img2 %s differ in shape" % (tuple(shape), tuple(img2.shape)))
        Cc, H, W = a.shape
        dev = a.device
        out = torch.empty(1, dtype=torch.float32, device=dev)
        # value pass now (it leaves the three partial-derivative maps), the adjoint correlation in backward() times the incoming gradient
        maps = torch.empty((3, Cc, H, W), dtype=torch.float32, device=dev) if img1.requires_grad else None
        sc = torch.empty(4096 + 4 * Cc * ((H + 31) // 32) * ((W + 31) // 32), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.hsr_loss_ssim_value(Cc, H, W, a.data_ptr(), b.data_ptr(), out.data_ptr(), None if maps is None else maps.data_ptr(),
                                          sc.data_ptr(), sc.numel(), _stream(dev))
        if rc < 0:
            _glue._fail(rc, "hsr_loss_ssim_value")
        ctx.want = maps is not None
        if ctx.want:
            ctx.save_for_backward(a, b, maps)
            ctx.meta = (Cc, H, W, tuple(shape))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if not ctx.want or g is None:
            return None, None
        a, b, maps = ctx.saved_tensors
        Cc, H, W, shape = ctx.meta
        dev = a.device
        gg = g.to(device=dev, dtype=torch.float32).contiguous()
        grad = torch.empty_like(a)
        with torch.cuda.device(dev):
            rc = _lib.hsr_loss_ssim_grad(Cc, H, W, a.data_ptr(), b.data_ptr(), maps.data_ptr(), gg.data_ptr(), grad.data_ptr(), _stream(dev))
        if rc < 0:
            _glue._fail(rc, "hsr_loss_ssim_grad")
        return grad.view(shape), None


class _TreeCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, level_sizes, weights, ignore_index):
        z = _chw(logits, "logits")
        K, H, W = z.shape
        dev = z.device
        L = len(level_sizes)
        lab = labels.reshape(-1, H, W)
        if lab.shape[0] < L:
            raise RuntimeError("hsr_utils.losses: %d label planes for %d levels" % (lab.shape[0], L))
        lab = lab[:L].to(device=dev, dtype=torch.int64).contiguous()   # the reference calls .long()
        sizes = (_ci * L)(*[int(s) for s in level_sizes])
        w = None if weights is None else (C.c_float * L)(*[float(x) for x in weights])
        # value now, gradient when (and if) autograd asks for it — the gradient pass multiplies by the incoming gradient itself, so
        # no K x H x W gradient is stashed and none is multiplied by `g` afterwards (include/hsr_losses.h, hsr_loss_tree_ce_value / _grad)
        out = torch.empty(L, dtype=torch.float32, device=dev)
        inv = torch.empty(L, dtype=torch.float32, device=dev)
        sc = torch.empty(int(_lib.hsr_loss_tree_ce_scratch_bytes(H, W)), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.hsr_loss_tree_ce_value(K, H, W, L, sizes, z.data_ptr(), lab.data_ptr(), int(ignore_index), out.data_ptr(),
                                             inv.data_ptr(), sc.data_ptr(), sc.numel(), _stream(dev))
        if rc < 0:
            _glue._fail(rc, "hsr_loss_tree_ce_value")
        ctx.want = bool(logits.requires_grad)
        if ctx.want:
            ctx.save_for_backward(z, lab, inv)
            ctx.meta = (K, H, W, L, sizes, w, int(ignore_index), tuple(logits.shape))
        ctx.mark_non_differentiable(out)
        if weights is None:
            return out.sum(), out
        key = (tuple(float(x) for x in weights), dev)
        if key not in _weight_cache:
            _weight_cache[key] = torch.tensor(key[0], dtype=torch.float32, device=dev)
        return (out * _weight_cache[key]).sum(), out

    @staticmethod
    def backward(ctx, g_total, g_levels):
        # the gradient of the weighted sum; per-level outputs are reported values (no gradient path)
        if not ctx.want or g_total is None:
            return None, None, None, None, None
        z, lab, inv = ctx.saved_tensors
        K, H, W, L, sizes, w, ignore_index, shape = ctx.meta
        dev = z.device
        g = g_total.to(device=dev, dtype=torch.float32).contiguous()
        grad = torch.empty_like(z)
        with torch.cuda.device(dev):
            rc = _lib.hsr_loss_tree_ce_grad(K, H, W, L, sizes, w, z.data_ptr(), lab.data_ptr(), ignore_index, inv.data_ptr(), g.data_ptr(),
                                            None, None, 0.0, grad.data_ptr(), _stream(dev))
        if rc < 0:
            _glue._fail(rc, "hsr_loss_tree_ce_grad")
        return grad.view(shape), None, None, None, None


class _WeightedSum(torch.autograd.Function):
    """constant + sum_i w_i * term_i over 0-dim device tensors with Python-float weights, as ONE autograd node.  The reference composes its
    loss with Python arithmetic on 0-dim tensors (scripts/hierslam.py:1003-1016: `0.8 * l1 + 0.2 * (1.0 - ssim)`, the weighted dictionary
    sum): every `*`, `+`, `1.0 -` there is a kernel launch forward and another backward — sixteen ~4 us launches around a mapping iteration's
    three loss heads.  Here: stack + dot forward, one scale backward (the terms' gradients are views of it)."""

    @staticmethod
    def forward(ctx, wvec, *terms):
        ctx.save_for_backward(wvec)
        return torch.dot(torch.stack(terms), wvec)

    @staticmethod
    def backward(ctx, g):
        (wvec,) = ctx.saved_tensors
        return (None,) + tuple((wvec * g).unbind(0))


def weighted_sum(terms, weights, constant=0.0):
    """constant + sum_i weights[i] * terms[i]; terms: 0-dim float32 tensors on one device (loss heads), weights / constant: Python floats."""
    terms = list(terms)
    if len(terms) != len(weights) or not terms:
        raise RuntimeError("hsr_utils.losses.weighted_sum: %d terms for %d weights" % (len(terms), len(weights)))
    dev = terms[0].device
    w = [float(x) for x in weights]
    if constant:
        key1 = ("one", dev)
        if key1 not in _weight_cache:
            _weight_cache[key1] = torch.ones((), dtype=torch.float32, device=dev)
        terms.append(_weight_cache[key1])
        w.append(float(constant))
    key = (tuple(w), dev)
    if key not in _weight_cache:
        _weight_cache[key] = torch.tensor(w, dtype=torch.float32, device=dev)
    return _WeightedSum.apply(_weight_cache[key], *terms)


class _TrackingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, gt_im, depth, gt_depth, silhouette, sil_thres, use_sil, w_depth, w_im, reduction):
        d, gd = _chw(depth, "depth"), _chw(gt_depth.detach(), "gt_depth")
        if im is None:      # depth term alone (the mapping branch)
            a = b = None
            Cc, (H, W) = 0, d.shape[-2:]
        else:
            a, b = _chw(im, "im"), _chw(gt_im.detach(), "gt_im")
            Cc, H, W = a.shape
        if (a is not None and b.shape != a.shape) or d.numel() != H * W or gd.numel() != H * W:
            raise RuntimeError("hsr_utils.losses: tracking_loss wants im / gt_im [C,H,W] and depth / gt_depth [1,H,W] of one size")
        s = None
        if use_sil:
            s = _chw(silhouette.detach(), "silhouette")
            if s.numel() != H * W:
                raise RuntimeError("hsr_utils.losses: silhouette must be [1,H,W] like the depth map")
        dev = d.device
        out = torch.empty(4, dtype=torch.float32, device=dev)
        sc = torch.empty(int(_lib.hsr_loss_tracking_scratch_bytes(H, W)), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.hsr_loss_tracking_value(Cc, H, W, None if a is None else a.data_ptr(), None if b is None else b.data_ptr(), d.data_ptr(),
                                              gd.data_ptr(), None if s is None else s.data_ptr(), float(sil_thres), int(bool(use_sil)), int(reduction),
                                              float(w_depth), float(w_im), out.data_ptr(), sc.data_ptr(), sc.numel(), _stream(dev))
        if rc < 0:
            _glue._fail(rc, "hsr_loss_tracking_value")
        ctx.want = (bool(im is not None and im.requires_grad), bool(depth.requires_grad))
        if any(ctx.want):
            e = torch.empty(0, device=dev)
            ctx.save_for_backward(a if a is not None else e, b if b is not None else e, d, gd, s if s is not None else e, out)
            ctx.meta = (Cc, H, W, float(sil_thres), int(bool(use_sil)), float(w_depth), float(w_im), None if im is None else tuple(im.shape),
                        tuple(depth.shape), int(reduction))
        parts = out[:2].detach()
        ctx.mark_non_differentiable(parts)
        return out[2], parts

    @staticmethod
    def backward(ctx, g, _g_parts):
        if g is None or not any(ctx.want):
            return (None,) * 10
        a, b, d, gd, s, out = ctx.saved_tensors
        Cc, H, W, sil_thres, use_sil, w_depth, w_im, shape_im, shape_d, reduction = ctx.meta
        dev = d.device
        gg = g.to(device=dev, dtype=torch.float32).contiguous()
        d_im = torch.empty_like(a) if ctx.want[0] else None
        d_d = torch.empty_like(d) if ctx.want[1] else None
        inv_ptr = out.data_ptr() + 12 if reduction == MEAN else None     # &out4[3]: 1 / selected pixels
        with torch.cuda.device(dev):
            rc = _lib.hsr_loss_tracking_grad(Cc, H, W, a.data_ptr() if Cc else None, b.data_ptr() if Cc else None, d.data_ptr(), gd.data_ptr(),
                                             s.data_ptr() if use_sil else None, sil_thres, use_sil, w_depth, w_im, gg.data_ptr(), inv_ptr,
                                             None if d_im is None else d_im.data_ptr(), None if d_d is None else d_d.data_ptr(), _stream(dev))
        if rc < 0:
            _glue._fail(rc, "hsr_loss_tracking_grad")
        return (None if d_im is None else d_im.view(shape_im), None, None if d_d is None else d_d.view(shape_d), None, None, None, None, None, None, None)


def tracking_loss(im, gt_im, depth, gt_depth, silhouette=None, sil_thres=0.99, use_sil_for_loss=True, loss_weights=None, return_parts=False):
    """The tracking loss of the reference's get_loss* with its shipped tracking settings (scripts/hierslam.py:903-937, :1003-1016:
    use_l1, ignore_outlier_depth_loss=False):
        mask = (gt_depth > 0) & ~isnan(depth) & (silhouette > sil_thres)     [the last factor if use_sil_for_loss]
        loss = loss_weights['depth'] * |gt_depth - depth|[mask].sum() + loss_weights['im'] * |gt_im - im|[tiled mask].sum()
    as one autograd node: one pass for the value, one for both gradients.  loss_weights: dict with 'im' and 'depth' (default the
    reference's tracking weights, im 0.5 / depth 1.0).  return_parts: also [depth sum, colour sum] (values, no gradient path).
    The outlier-rejecting variant (a global median of the depth error) is not fused: compose masked_l1 with a torch mask for it."""
    lw = loss_weights or {"im": 0.5, "depth": 1.0}
    if use_sil_for_loss and silhouette is None:
        raise RuntimeError("hsr_utils.losses: tracking_loss with use_sil_for_loss needs the rendered silhouette / final opacity map")
    total, parts = _TrackingLoss.apply(im, gt_im, depth, gt_depth, silhouette, float(sil_thres), bool(use_sil_for_loss), float(lw["depth"]),
                                       float(lw["im"]), SUM)
    return (total, parts) if return_parts else total


def mapping_depth_loss(depth, gt_depth):
    """The depth term of the mapping branch of get_loss* (scripts/hierslam.py:905-927 with the shipped mapping settings: no silhouette
    mask, no outlier rejection):  torch.abs(gt_depth - depth)[(gt_depth > 0) & ~isnan(depth)].mean()  — mask, count and mean in one pass,
    the gradient in a second when autograd asks (no mask tensor, no count pre-pass)."""
    total, _parts = _TrackingLoss.apply(None, None, depth, gt_depth, None, 0.0, False, 1.0, 0.0, MEAN)
    return total


def l1_loss_v1(x, y):
    """torch.abs(x - y).mean() (utils/slam_helpers.py:5-6); gradient to x."""
    return _L1.apply(x, y, None, MEAN)


def masked_l1(pred, gt, mask, reduction="sum"):
    """torch.abs(gt - pred)[mask].sum() / .mean() with a [H,W] or [1,H,W] mask tiled over the channels
    (scripts/hierslam.py:921-937); gradient to pred.  mask=None selects everything."""
    return _L1.apply(pred, gt, mask, {"sum": SUM, "mean": MEAN}[reduction])


def calc_ssim(img1, img2, window_size=11, size_average=True):
    """utils/slam_external.py:66-97 for its defaults (11x11 window, mean over the map); gradient to img1."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("hsr_utils.losses.calc_ssim: only window_size=11, size_average=True (what scripts/hierslam.py uses)")
    return _SSIM.apply(img1, img2)


def mapping_image_loss(im, gt):
    """0.8 * l1_loss_v1(im, gt) + 0.2 * (1.0 - calc_ssim(im, gt))   (scripts/hierslam.py:939)"""
    return weighted_sum((l1_loss_v1(im, gt), calc_ssim(im, gt)), (0.8, -0.2), constant=0.2)


def tree_cross_entropy(im_semantic, labels, num_semantic, weights=None, ignore_index=-100, return_levels=False):
    """sum_l w_l * CrossEntropyLoss()(im_semantic[level l channels] as [H*W, n_l], labels[l].view(-1).long())
    (scripts/hierslam.py:963-974 with transfer_tree_rendered_labelmap, :91-111).  `num_semantic` lists the classes per
    level; labels is [>= len(num_semantic), H, W].  Gradient to im_semantic."""
    total, levels = _TreeCE.apply(im_semantic, labels, tuple(num_semantic), None if weights is None else tuple(weights), ignore_index)
    return (total, levels) if return_levels else total


def cross_entropy_planar(logits, labels, ignore_index=-100):
    """CrossEntropyLoss()(logits.permute(1,2,0).view(-1, C), labels.view(-1).long()) for [C,H,W] logits
    (flat classes: scripts/hierslam.py:947-954; leaf MLP output: :976-983)."""
    z = logits[0] if logits.dim() == 4 else logits
    return tree_cross_entropy(z, labels.reshape(1, z.shape[-2], z.shape[-1]), (z.shape[0],), None, ignore_index)


class _LeafMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sem, weight, bias, labels, ignore_index):
        z = _chw(sem, "im_semantic")
        K, H, W = z.shape
        dev = z.device
        w = _dev2(weight.reshape(weight.shape[0], -1), "weight")
        C = w.shape[0]
        if w.shape[1] != K or bias.numel() != C:
            raise RuntimeError("hsr_utils.losses: weight %s / bias %s do not match %d input channels" % (tuple(weight.shape), tuple(bias.shape), K))
        b = _dev2(bias.reshape(-1), "bias")
        lab = labels.reshape(H, W).to(device=dev, dtype=torch.int64).contiguous()
        out = torch.empty(1, dtype=torch.float32, device=dev)
        need = (sem.requires_grad, weight.requires_grad, bias.requires_grad)
        d_sem = torch.empty_like(z) if need[0] else None
        d_w = torch.empty_like(w) if (need[1] or need[2]) else None
        d_b = torch.empty_like(b) if (need[1] or need[2]) else None
        sc = _scratch(K, H, W, dev)
        with torch.cuda.device(dev):
            rc = _lib.hsr_loss_leaf_mlp_ce(K, C, H, W, z.data_ptr(), w.data_ptr(), b.data_ptr(), lab.data_ptr(), int(ignore_index),
                                           out.data_ptr(), None if d_sem is None else d_sem.data_ptr(), None if d_w is None else d_w.data_ptr(),
                                           None if d_b is None else d_b.data_ptr(), sc.data_ptr(), sc.numel(), _stream(dev))
        if rc < 0:
            _glue._fail(rc, "hsr_loss_leaf_mlp_ce")
        ctx.grads = (None if d_sem is None else d_sem.view(sem.shape), None if d_w is None else d_w.view(weight.shape),
                     None if d_b is None else d_b.view(bias.shape))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        ds, dw, db = ctx.grads
        return (None if ds is None else ds * g, None if dw is None else dw * g, None if db is None else db * g, None, None)


class _SemanticHeads(torch.autograd.Function):
    """losses['sem'] of get_loss_semantic_mlp once the leaf head is on (scripts/hierslam.py:963-983): w_tree * sum over the tree levels of the
    per-level cross-entropy + w_leaf * cross-entropy of the 1x1-conv leaf head — as ONE node over the semantic map.  Forward: the tree value
    pass and the fused leaf kernel (which stashes d leaf / d sem).  Backward: ONE pass writes d total / d sem — the tree gradient pass takes
    the leaf head's stashed gradient along (hsr_loss_tree_ce_grad, add_grad) instead of a `stash * g` pass and autograd's add of two
    K x H x W maps."""

    @staticmethod
    def forward(ctx, sem, weight, bias, labels, level_sizes, w_tree, w_leaf, ignore_index):
        z = _chw(sem, "im_semantic")
        K, H, W = z.shape
        dev = z.device
        L = len(level_sizes)
        lab = labels.reshape(-1, H, W)
        if lab.shape[0] < L + 1:
            raise RuntimeError("hsr_utils.losses: %d label planes for %d tree levels + the leaf level" % (lab.shape[0], L))
        lab = lab.to(device=dev, dtype=torch.int64).contiguous()
        tree_lab, leaf_lab = lab[:L], lab[-1]
        wt = _dev2(weight.reshape(weight.shape[0], -1), "weight")
        Cc = wt.shape[0]
        if wt.shape[1] != K or bias.numel() != Cc:
            raise RuntimeError("hsr_utils.losses: weight %s / bias %s do not match %d input channels" % (tuple(weight.shape), tuple(bias.shape), K))
        b = _dev2(bias.reshape(-1), "bias")
        sizes = (_ci * L)(*[int(s) for s in level_sizes])
        levels = torch.empty(L, dtype=torch.float32, device=dev)
        inv = torch.empty(L, dtype=torch.float32, device=dev)
        leaf = torch.empty(1, dtype=torch.float32, device=dev)
        need = (sem.requires_grad, weight.requires_grad, bias.requires_grad)
        d_sem = torch.empty_like(z) if need[0] else None
        d_w = torch.empty_like(wt) if (need[1] or need[2]) else None
        d_b = torch.empty_like(b) if (need[1] or need[2]) else None
        sc1 = torch.empty(int(_lib.hsr_loss_tree_ce_scratch_bytes(H, W)), dtype=torch.uint8, device=dev)
        sc2 = _scratch(K, H, W, dev)
        with torch.cuda.device(dev):
            rc = _lib.hsr_loss_tree_ce_value(K, H, W, L, sizes, z.data_ptr(), tree_lab.data_ptr(), int(ignore_index), levels.data_ptr(),
                                             inv.data_ptr(), sc1.data_ptr(), sc1.numel(), _stream(dev))
            if rc < 0:
                _glue._fail(rc, "hsr_loss_tree_ce_value")
            rc = _lib.hsr_loss_leaf_mlp_ce(K, Cc, H, W, z.data_ptr(), wt.data_ptr(), b.data_ptr(), leaf_lab.data_ptr(), int(ignore_index),
                                           leaf.data_ptr(), None if d_sem is None else d_sem.data_ptr(), None if d_w is None else d_w.data_ptr(),
                                           None if d_b is None else d_b.data_ptr(), sc2.data_ptr(), sc2.numel(), _stream(dev))
            if rc < 0:
                _glue._fail(rc, "hsr_loss_leaf_mlp_ce")
        ctx.want = need
        if need[0]:
            ctx.save_for_backward(z, tree_lab, inv)
        ctx.meta = (K, H, W, L, sizes, float(w_tree), float(w_leaf), int(ignore_index), tuple(sem.shape))
        ctx.stash = (d_sem, None if d_w is None else d_w.view(weight.shape), None if d_b is None else d_b.view(bias.shape))
        ctx.mark_non_differentiable(levels)
        total = levels.sum() * float(w_tree) + leaf[0] * float(w_leaf)
        return total, levels, leaf[0].detach()

    @staticmethod
    def backward(ctx, g, _g_levels, _g_leaf):
        K, H, W, L, sizes, w_tree, w_leaf, ignore_index, shape = ctx.meta
        d_sem, d_w, d_b = ctx.stash
        if g is None:
            return (None,) * 8
        out_sem = None
        if ctx.want[0]:
            z, tree_lab, inv = ctx.saved_tensors
            dev = z.device
            gg = g.to(device=dev, dtype=torch.float32).contiguous()
            wl = (C.c_float * L)(*([w_tree] * L))
            grad = torch.empty_like(z)
            with torch.cuda.device(dev):
                rc = _lib.hsr_loss_tree_ce_grad(K, H, W, L, sizes, wl, z.data_ptr(), tree_lab.data_ptr(), ignore_index, inv.data_ptr(),
                                                gg.data_ptr(), d_sem.data_ptr(), gg.data_ptr(), w_leaf, grad.data_ptr(), _stream(dev))
            if rc < 0:
                _glue._fail(rc, "hsr_loss_tree_ce_grad")
            out_sem = grad.view(shape)
        gl = g * w_leaf
        return (out_sem, None if (d_w is None or not ctx.want[1]) else d_w * gl, None if (d_b is None or not ctx.want[2]) else d_b * gl,
                None, None, None, None, None)


def semantic_loss_mlp(im_semantic, labels, num_semantic, mlp, weight_sem=(1.0, 1.0), ignore_index=-100, return_parts=False):
    """losses['sem'] of the reference's get_loss_semantic_mlp with the leaf head on (scripts/hierslam.py:963-983):
        weight_sem[0] * sum_l CrossEntropyLoss(level l of the tree) + weight_sem[1] * CrossEntropyLoss(MLP_func(im_semantic), leaf labels)
    labels: [len(num_semantic) + 1, H, W], the last plane the leaf labels (curr_data['semantic_label_gt'] as the reference holds it).
    One autograd node; gradients to im_semantic, mlp.weight, mlp.bias.  Heads wider than the fused leaf kernel takes fall back to the two
    separate heads.  return_parts: also the per-level tree losses and the leaf loss (values, no gradient path)."""
    weight, bias = (mlp.weight, mlp.bias) if hasattr(mlp, "weight") else mlp
    K = im_semantic.shape[-3]
    if K > LEAF_MAX_K or weight.shape[0] > LEAF_MAX_C:
        L = len(num_semantic)
        lab = labels.reshape(-1, *im_semantic.shape[-2:])
        tree, levels = tree_cross_entropy(im_semantic, lab[:L], num_semantic, None, ignore_index, return_levels=True)
        leaf = leaf_mlp_cross_entropy(im_semantic, mlp, lab[-1], ignore_index)
        total = weighted_sum((tree, leaf), (float(weight_sem[0]), float(weight_sem[1])))
        return (total, levels, leaf.detach()) if return_parts else total
    total, levels, leaf = _SemanticHeads.apply(im_semantic, weight, bias, labels, tuple(num_semantic), float(weight_sem[0]), float(weight_sem[1]),
                                               ignore_index)
    return (total, levels, leaf) if return_parts else total


def _dev2(t, what):
    if not t.is_cuda:
        raise RuntimeError("hsr_utils.losses: %s must live on a HIP device (got %s); there is no CPU path" % (what, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("hsr_utils.losses: %s must be float32 (got %s)" % (what, t.dtype))
    return t.contiguous()


def leaf_mlp_cross_entropy(im_semantic, mlp, labels, ignore_index=-100):
    """CrossEntropyLoss()(MLP_func(im_semantic.unsqueeze(0)) as [H*W, C], labels.view(-1).long()) with MLP_func a
    torch.nn.Conv2d(K, C, kernel_size=1) (scripts/hierslam.py:1756, :976-983), value and gradients to im_semantic,
    mlp.weight and mlp.bias from ONE kernel that never materialises the [C,H,W] logits.  `mlp` may also be a
    (weight, bias) pair.  Wider heads (K > 31 or C > 128) run the conv in torch and the fused planar cross-entropy."""
    weight, bias = (mlp.weight, mlp.bias) if hasattr(mlp, "weight") else mlp
    K = im_semantic.shape[-3]
    if K > LEAF_MAX_K or weight.shape[0] > LEAF_MAX_C:
        logits = torch.nn.functional.conv2d(im_semantic.reshape(1, K, *im_semantic.shape[-2:]), weight.reshape(weight.shape[0], K, 1, 1), bias)
        return cross_entropy_planar(logits, labels, ignore_index)
    return _LeafMLP.apply(im_semantic, weight, bias, labels, ignore_index)
