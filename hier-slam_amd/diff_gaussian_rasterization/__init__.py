"""diff_gaussian_rasterization — drop-in Python surface of Hier-SLAM's differentiable Gaussian rasterizer,
backed by the MI355X-native HIP library (libhsr_rast.so) instead of the reference's CUDA extension.

Same public names, argument names, return tuples and error behaviour as the reference package
(hierslam-diff-gaussian-rasterization-w-depth/diff_gaussian_rasterization/__init__.py):

    GaussianRasterizationSettings            (:161-173)
    GaussianRasterizer                       (:175-227)  -> (color, radii, depth, median_depth, final_opacity, mask)
    GaussianRasterizer_semantic              (:377-430)  -> (color, radii, semantic, depth, median_depth, final_opacity)
    rasterize_gaussians / rasterize_gaussians_semantic   (:21-42, :231-252)

so `from diff_gaussian_rasterization import GaussianRasterizer as Renderer` in scripts/hierslam.py:53-54,
utils/recon_helpers.py:2 and utils/eval_helpers.py:21-22 keeps working unchanged.  The number of semantic
channels K is taken from `semantics_precomp.shape[1]` at run time (the reference bakes NUM_SEMANTIC into the
build, config.h:18).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "GaussianRasterizer_semantic",
           "rasterize_gaussians", "rasterize_gaussians_semantic", "set_async_forward", "set_semantic_alpha"]


def set_async_forward(on):
    """Opt-in: a forward that a backward will follow returns before the device has counted num_rendered (the reference — and the
    default here — waits for that 4-byte read-back in every frame).  See diff_gaussian_rasterization/_C.py for the one condition
    under which this raises (num_rendered more than doubling between two frames of the same size).  Errors the device finds during such a
    forward — that overflow, a `prefiltered` violation — surface when the count is resolved (in the backward), not at the forward call.
    Returns the previous setting."""
    return _C.set_async_forward(on)


def set_semantic_alpha(mode):
    """'reference' (default) or 'exact': whether the semantic loss reaches alpha (opacity, covariance, position).  The reference's does
    not (its backward reads an unwritten staging array there); 'exact' adds the intended term.  See _C.set_semantic_alpha."""
    _C.set_semantic_alpha(mode)


import threading

_caller = threading.local()   # the grad mode of the thread that called rasterize_gaussians*(): see _Rasterize.forward

class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _snapshot(args):
    """CPU deep copy of the call arguments, taken before the call in debug mode (reference :17-19)."""
    # numpy scalars (a tanfov computed with numpy) become Python numbers, so the dump loads with torch.load(weights_only=True)
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else (a.item() if hasattr(a, "item") else a) for a in args)


def _call(fn, args, debug, dump_name, when, **options):
    """Runs one _C entry point; in debug mode a failure first dumps the inputs (reference :82-90, :294-301)."""
    if not debug:
        return fn(*args, **options)
    saved = _snapshot(args)
    try:
        return fn(*args, **options)
    except Exception:
        torch.save(saved, dump_name)
        print("\nAn error occured in %s. Please forward %s for debugging." % (when, dump_name))
        raise


_zero_maps = {}   # (device, shape) -> a float32 zero tensor that is only ever READ (see _Rasterize.backward)


def _zeros_for(shape, device):
    key = (device, tuple(shape))
    z = _zero_maps.get(key)
    if z is None:
        z = _zero_maps[key] = torch.zeros(tuple(shape), dtype=torch.float32, device=device)
    return z


class _Rasterize(torch.autograd.Function):
    """One autograd node for both variants.  Inputs (semantic variant):
    means3D, means2D, sh, colors_precomp, semantics_precomp, opacities, scales, rotations, cov3Ds_precomp, settings;
    the plain variant passes semantics_precomp=None and gets `mask` in place of `semantic`."""

    @staticmethod
    def forward(ctx, semantic, means3D, means2D, sh, colors_precomp, semantics_precomp, opacities, scales, rotations,
                cov3Ds_precomp, rs):
        common = (opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix, rs.projmatrix,
                  rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered,
                  rs.debug)
        # non-blocking forward (opt-in, _C.set_async_forward): only where a backward will follow and resolve num_rendered
        # (needs_input_grad reflects the inputs' requires_grad whatever the caller's grad mode: under torch.no_grad() no backward can
        # follow — and inside forward() the grad mode is always off, so the wrappers below note the caller's)
        ahead = dict(run_ahead=True) if (_C._async_forward and getattr(_caller, "grad_enabled", True) and any(ctx.needs_input_grad)) else {}
        if semantic:
            args = (rs.bg, means3D, colors_precomp, semantics_precomp) + common
            (num_rendered, color, aux, depth, median_depth, final_opacity, radii, geom, binning, img) = _call(
                _C.rasterize_gaussians_semantic, args, rs.debug, "snapshot_fw.dump", "forward", **ahead)
        else:
            args = (rs.bg, means3D, colors_precomp) + common
            (num_rendered, color, depth, median_depth, final_opacity, aux, radii, geom, binning, img) = _call(
                _C.rasterize_gaussians, args, rs.debug, "snapshot_fw.dump", "forward", **ahead)
        ctx.semantic = semantic
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        # Outputs the loss does not use arrive in backward() as None instead of freshly zero-filled maps (autograd's default fills
        # one per unused output per step: a tracking iteration that looks at depth and colour only would fill — and the tile kernel
        # read — a K x H x W map of zeros every time).  backward() hands the library a cached, never-written zero map in their place.
        ctx.set_materialize_grads(False)
        ctx.map_shapes = (tuple(color.shape), tuple(aux.shape), tuple(depth.shape), tuple(median_depth.shape), tuple(final_opacity.shape))
        ctx.save_for_backward(colors_precomp, semantics_precomp if semantic else torch.empty(0), means3D, scales, rotations,
                              cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        if semantic:
            return color, radii, aux, depth, median_depth, final_opacity
        return color, radii, depth, median_depth, final_opacity, aux

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, g2, g3, g4, g5):
        rs = ctx.raster_settings
        (colors_precomp, semantics_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning,
         img) = ctx.saved_tensors
        if ctx.semantic:
            grad_sem, grad_depth, grad_median, grad_opacity = g2, g3, g4, g5
        else:
            grad_depth, grad_median, grad_opacity = g2, g3, g4  # g5 = grad of mask: ignored, like the reference (:101)
            grad_sem = None
        dev = means3D.device
        sh_color, sh_aux, sh_depth, sh_median, sh_opacity = ctx.map_shapes
        if grad_color is None:
            grad_color = _zeros_for(sh_color, dev)
        if grad_depth is None:
            grad_depth = _zeros_for(sh_depth, dev)
        if grad_median is None:
            grad_median = _zeros_for(sh_median, dev)
        if grad_opacity is None:
            grad_opacity = _zeros_for(sh_opacity, dev)
        if ctx.semantic and grad_sem is None:
            grad_sem = _zeros_for(sh_aux, dev)
        tail = (sh, rs.sh_degree, rs.campos, geom, ctx.num_rendered, binning, img, rs.debug)
        head = (rs.bg, means3D, radii, colors_precomp)
        mid = (scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
               grad_color)
        # Per-call options (arguments, not module state: autograd runs one backward thread per device):
        #  want_cov3D_grad  cov3Ds_precomp needs a gradient;
        #  geometry_only    only means3D / means2D want one (a tracking iteration optimises the camera pose alone,
        #                   scripts/hierslam.py:1683-1860): the library then forms the geometry sums only — a third of the atomic
        #                   traffic, no semantic upstream gradients read.
        opts = dict(want_cov3D_grad=bool(ctx.needs_input_grad[9]),
                    geometry_only=not any(ctx.needs_input_grad[i] for i in (3, 4, 5, 6, 7, 8, 9)))
        if ctx.semantic:
            args = head + (semantics_precomp,) + mid + (grad_sem, grad_depth, grad_median, grad_opacity) + tail
            (g_means2D, g_colors, g_sem, g_opac, g_means3D, g_cov3D, g_sh, g_scales, g_rot) = _call(
                _C.rasterize_gaussians_backward_semantic, args, rs.debug, "snapshot_bw.dump", "backward", **opts)
        else:
            args = head + mid + (grad_depth, grad_median, grad_opacity) + tail
            (g_means2D, g_colors, g_opac, g_means3D, g_cov3D, g_sh, g_scales, g_rot) = _call(
                _C.rasterize_gaussians_backward, args, rs.debug, "snapshot_bw.dump", "backward", **opts)
            g_sem = None
        # one gradient per forward input: (semantic flag, means3D, means2D, sh, colors, semantics, opacities,
        # scales, rotations, cov3Ds_precomp, settings)
        return (None, g_means3D, g_means2D, g_sh, g_colors, g_sem, g_opac, g_scales, g_rot, g_cov3D, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    _caller.grad_enabled = torch.is_grad_enabled()
    return _Rasterize.apply(False, means3D, means2D, sh, colors_precomp, None, opacities, scales, rotations,
                            cov3Ds_precomp, raster_settings)


def rasterize_gaussians_semantic(means3D, means2D, sh, colors_precomp, semantics_precomp, opacities, scales, rotations,
                                 cov3Ds_precomp, raster_settings):
    _caller.grad_enabled = torch.is_grad_enabled()
    return _Rasterize.apply(True, means3D, means2D, sh, colors_precomp, semantics_precomp, opacities, scales, rotations,
                            cov3Ds_precomp, raster_settings)


def _empty():
    return torch.Tensor([])


def _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp):
    # messages kept verbatim from the reference (:195-199), typo included, for callers that match on them
    if (shs is None) == (colors_precomp is None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    has_sr = scales is not None or rotations is not None
    if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')


class _RasterizerBase(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: view-space z > 0.2 (reference :180-189)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)


class GaussianRasterizer(_RasterizerBase):
    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        e = _empty
        return rasterize_gaussians(
            means3D, means2D,
            e() if shs is None else shs,
            e() if colors_precomp is None else colors_precomp,
            opacities,
            e() if scales is None else scales,
            e() if rotations is None else rotations,
            e() if cov3D_precomp is None else cov3D_precomp,
            self.raster_settings)


class GaussianRasterizer_semantic(_RasterizerBase):
    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, semantics_precomp=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        e = _empty
        return rasterize_gaussians_semantic(
            means3D, means2D,
            e() if shs is None else shs,
            e() if colors_precomp is None else colors_precomp,
            e() if semantics_precomp is None else semantics_precomp,
            opacities,
            e() if scales is None else scales,
            e() if rotations is None else rotations,
            e() if cov3D_precomp is None else cov3D_precomp,
            self.raster_settings)
