"""`diff_gaussian_rasterization._C` — torch-facing glue over libhsr_rast.so (the gfx950 HIP library).

Mirrors the five entry points the reference exports from its pybind module (ext.cpp:15-23,
rasterize_points.h:18-125) with the same argument order and return tuples, so
`diff_gaussian_rasterization/__init__.py` — and through it scripts/hierslam.py — binds to it unchanged:

    rasterize_gaussians, rasterize_gaussians_backward, mark_visible,
    rasterize_gaussians_semantic, rasterize_gaussians_backward_semantic

The glue only allocates tensors and hands raw device pointers to the C ABI declared in
include/hsr_rasterizer.h (ctypes; no torch types cross the boundary).  There is NO fallback path: if the
shared library is missing or the tensors are not on a HIP device, these functions raise.
"""
import ctypes as C
import itertools
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("HSR_RAST_LIB", os.path.join(os.path.dirname(_HERE), "libhsr_rast.so"))

NUM_CHANNELS = 3  # reference config.h:15


class _HsrBuffer(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("capacity", C.c_size_t), ("grow", C.c_void_p), ("user", C.c_void_p)]


_GROW_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)


class _Ticket(C.Structure):
    """hsr_ticket (include/hsr_rasterizer.h): a forward call that returned before num_rendered was known"""
    _fields_ = [("seq", C.c_uint32), ("device", C.c_int32), ("slot", C.c_void_p), ("binning_base", C.c_void_p),
                ("binning_capacity", C.c_size_t), ("prefiltered", C.c_int32), ("rendered", C.c_int32)]


HSR_PENDING = -100
HSR_ERR_BUFFER_TOO_SMALL = -2


class _StateLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "geom_depths", "geom_means2D", "geom_conic_opacity", "geom_cov3D", "geom_rgb", "geom_clamped",
        "geom_tiles_touched", "geom_point_offsets", "geom_radii",
        "bin_keys_unsorted", "bin_keys", "bin_vals_unsorted", "bin_vals",
        "img_ranges", "img_final_T", "img_n_contrib", "img_median_pos")]


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            "diff_gaussian_rasterization: HIP library not found at %s — build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C hier-slam_amd/csrc`. "
            "There is no CPU fallback." % _LIB_PATH)
    lib = C.CDLL(_LIB_PATH)
    vp, ci, cf, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    bp = C.POINTER(_HsrBuffer)
    lib.hsr_required_geometry_bytes.restype = sz
    lib.hsr_required_geometry_bytes.argtypes = [ci]
    lib.hsr_required_image_bytes.restype = sz
    lib.hsr_required_image_bytes.argtypes = [ci, ci]
    lib.hsr_required_binning_bytes.restype = sz
    lib.hsr_required_binning_bytes.argtypes = [ci]
    lib.hsr_last_error.restype = C.c_char_p
    lib.hsr_version.restype = C.c_char_p
    lib.hsr_mark_visible.restype = ci
    lib.hsr_mark_visible.argtypes = [ci, vp, vp, vp, vp, vp]
    lib.hsr_forward.restype = ci
    lib.hsr_forward.argtypes = [bp, bp, bp, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, cf, cf, ci,
                                vp, vp, vp, vp, vp, vp, ci, vp]
    lib.hsr_forward_semantic.restype = ci
    lib.hsr_forward_semantic.argtypes = [bp, bp, bp, ci, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp,
                                         cf, cf, ci, vp, vp, vp, vp, vp, vp, ci, vp]
    lib.hsr_backward.restype = ci
    lib.hsr_backward.argtypes = [ci, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp, vp,
                                 vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, ci, vp]
    lib.hsr_set_backward_mode.restype = ci
    lib.hsr_set_backward_mode.argtypes = [ci]
    lib.hsr_backward_scratch_bytes.restype = sz
    lib.hsr_backward_scratch_bytes.argtypes = [ci, ci, ci]
    lib.hsr_backward_semantic.restype = ci
    lib.hsr_backward_semantic.argtypes = [ci, ci, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, cf, cf,
                                          vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz,
                                          ci, vp]
    lib.hsr_stage_name.restype = C.c_char_p
    lib.hsr_stage_name.argtypes = [ci]
    lib.hsr_profile_enable.restype = ci
    lib.hsr_profile_enable.argtypes = [ci]
    lib.hsr_get_backward_mode.restype = ci
    lib.hsr_get_backward_mode.argtypes = []
    lib.hsr_set_semantic_alpha_mode.restype = ci
    lib.hsr_set_semantic_alpha_mode.argtypes = [ci]
    lib.hsr_get_semantic_alpha_mode.restype = ci
    lib.hsr_get_semantic_alpha_mode.argtypes = []
    lib.hsr_profile_host_wait_ms.restype = C.c_double
    lib.hsr_profile_host_wait_ms.argtypes = [ci]
    lib.hsr_profile_select.restype = ci
    lib.hsr_profile_select.argtypes = [C.c_uint]
    lib.hsr_profile_read.restype = ci
    lib.hsr_profile_read.argtypes = [vp, ci]
    lib.hsr_get_state_layout.restype = ci
    lib.hsr_get_state_layout.argtypes = [ci, ci, ci, ci, C.POINTER(_StateLayout)]
    lib.hsr_forward_arm_async.restype = ci
    lib.hsr_forward_arm_async.argtypes = [C.POINTER(_Ticket)]
    lib.hsr_forward_end.restype = ci
    lib.hsr_forward_end.argtypes = [C.POINTER(_Ticket), ci, vp]
    return lib


_lib = _load()

# The same glue as a compiled extension (csrc/hsr_torch_ext.cpp, built by csrc/build_torch_ext.py): used for the four
# rasterize entry points when present, because the interpreter work per call matters at 0.65 ms per render.  HSR_GLUE=ctypes
# forces the pure-Python path below (identical behaviour; tests run both).
_ext = None
if os.environ.get("HSR_GLUE", "") != "ctypes":
    try:
        from . import _hsr_torch as _ext  # noqa: F401
    except ImportError:
        _ext = None

# last num_rendered seen per (device, P, W, H): sizes the binning buffer up front so that the steady
# state needs no grow callback (the reference resizes through a callback every call,
# rasterize_points.cu:27-33)
class _HintMap(dict):
    """(device index, P, W, H) -> last num_rendered.  Lives in the extension when that is loaded."""

    def _e(self, key, v):
        return _ext.binning_hint(int(key[0]), int(key[1]), int(key[2]), int(key[3]), int(v))

    def __getitem__(self, key):
        if _ext is None:
            return dict.__getitem__(self, key)
        v = self._e(key, -1)
        if v < 0:
            raise KeyError(key)
        return v

    def __setitem__(self, key, v):
        if _ext is None:
            dict.__setitem__(self, key, v)
        else:
            self._e(key, v)

    def __delitem__(self, key):
        if _ext is None:
            dict.__delitem__(self, key)
        else:
            self._e(key, -2)

    def __contains__(self, key):
        return dict.__contains__(self, key) if _ext is None else self._e(key, -1) >= 0

    def get(self, key, default=None):
        return self[key] if key in self else default

    def clear(self):
        if _ext is None:
            dict.clear(self)
        else:
            self._e((0, 0, 0, 0), -3)

    def pop(self, key, *default):
        if key in self:
            v = self[key]
            del self[key]
            return v
        if default:
            return default[0]
        raise KeyError(key)


_binning_hint = _HintMap()

# Non-blocking forward (opt-in).  The reference's forward stalls host and stream on a 4-byte read-back of num_rendered in every frame
# (rasterizer_impl.cu:285 / :548); by default this glue also returns num_rendered as an int, i.e. waits for the device to have counted.
# set_async_forward(True): a forward whose inputs need gradients, on a (device, P, W, H) that has rendered before, enqueues ALL its
# kernels into a binning buffer sized for TWICE the previous count and returns at once; num_rendered comes back as a LazyRendered
# that resolves itself when it is first used as a number (the autograd backward does: by then the count has long been written).
# The price, and the reason this is not the default: should num_rendered more than double from one frame to the next, the buffer is
# too small, nothing could be rendered, the outputs of that call are all NaN, and resolving the count raises — the step has to be
# repeated (the next call's buffer is sized from the count that did not fit).
_async_forward = os.environ.get("HSR_ASYNC_FORWARD", "") == "1"


def set_async_forward(on):
    """enable / disable the non-blocking forward (see above); returns the previous setting"""
    global _async_forward
    prev, _async_forward = _async_forward, bool(on)
    return prev


class LazyRendered:
    """num_rendered of a forward call that ran ahead of the device: an int-like that fetches the count (and checks that it fitted
    the binning buffer) the first time it is used as a number"""
    __slots__ = ("_ticket", "_value", "_key", "_stream", "_error")

    def __init__(self, ticket, key, stream):
        self._ticket, self._value, self._key, self._stream, self._error = ticket, None, key, stream, None

    def ready(self):
        """has the device written the count yet? (never blocks)"""
        return self._value is not None or self._end(False) is not None

    def _end(self, block):
        if self._value is not None:
            return self._value
        if self._error is not None:
            raise RuntimeError(self._error)        # reported once already: the same answer to whoever asks again
        if isinstance(self._ticket, bytes):
            rc, rendered, err = _ext.forward_end(self._ticket, bool(block), int(self._stream))
        else:
            rc = _lib.hsr_forward_end(C.byref(self._ticket), int(bool(block)), self._stream)
            rendered, err = int(self._ticket.rendered), (_lib.hsr_last_error().decode() if rc < 0 else "")
        if rc == HSR_PENDING:
            return None
        if rc == HSR_ERR_BUFFER_TOO_SMALL:
            _binning_hint[self._key] = int(rendered)      # the next forward of this size gets room for it
            self._error = "diff_gaussian_rasterization (async forward): " + err
            raise RuntimeError(self._error)
        if rc < 0:
            self._error = "diff_gaussian_rasterization (async forward): hsr_forward_end failed (code %d): %s" % (rc, err)
            raise RuntimeError(self._error)
        self._value = int(rc)
        _binning_hint[self._key] = self._value
        return self._value

    def __int__(self):
        return self._end(True)

    __index__ = __int__

    def __bool__(self):
        return int(self) != 0

    def __sub__(self, other):
        return int(self) - other

    def __rsub__(self, other):
        return other - int(self)

    def __eq__(self, other):
        return int(self) == other

    def __ne__(self, other):
        return int(self) != other

    def __lt__(self, other):
        return int(self) < other

    def __le__(self, other):
        return int(self) <= other

    def __gt__(self, other):
        return int(self) > other

    def __ge__(self, other):
        return int(self) >= other

    def __hash__(self):
        return hash(int(self))

    def __repr__(self):
        return "LazyRendered(%s)" % ("pending" if self._value is None else self._value)

    def __format__(self, spec):
        return format(int(self), spec)

    def __add__(self, other):
        return int(self) + other

    __radd__ = __add__

    def __mul__(self, other):
        return int(self) * other

    __rmul__ = __mul__

    def __truediv__(self, other):
        return int(self) / other

    def __floordiv__(self, other):
        return int(self) // other


# The newest unresolved count of every (device, P, W, H): a forward that ran ahead and was never followed by a backward (a
# visualisation render outside no_grad) would otherwise never report an overflow — its outputs are NaN by then — and never correct the
# binning hint, so every later frame of that size would overflow again.  The next run-ahead forward of the same key
# resolves it first: an overflow of frame i is raised, loudly, at the forward of frame i + 1 at the latest.
_unresolved = {}


def _resolve_previous(key):
    prev = _unresolved.pop(key, None)
    if prev is not None and prev._value is None and prev._error is None:
        prev._end(True)


# Gradient sink (hsr_utils/parallel.py GradientExchange): a callable (name, shape, device) -> tensor | None that may hand the
# backward a PRE-ALLOCATED output tensor — a fresh view of a communication bucket — for a named gradient, so that the
# gradient is written where the all-reduce will read it and nothing is copied afterwards.  Names: "raster.means3D",
# "raster.colors_precomp", "raster.semantics_precomp", "raster.opacities", "raster.scales", "raster.rotations" here;
# "params.means3D", "params.unnorm_rotations", "params.logit_opacities", "params.log_scales" in hsr_utils/slam_helpers.py.
# None (default): every gradient is a fresh allocation, as in the reference (rasterize_points.cu:378-388).
_gradient_sink = None


def set_gradient_sink(fn):
    """install (or, with None, remove) the gradient sink; returns the previous one"""
    global _gradient_sink
    prev, _gradient_sink = _gradient_sink, fn
    return prev


def _from_sink(name, shape, dev):
    if _gradient_sink is None:
        return None
    t = _gradient_sink(name, tuple(int(x) for x in shape), dev)
    if t is None:
        return None
    if tuple(t.shape) != tuple(shape) or t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
        raise RuntimeError("gradient sink returned a tensor of the wrong shape / dtype / device / layout for %s" % name)
    return t

def _rows_or_legacy():
    return int(_lib.hsr_get_backward_mode()) != 0


def set_backward_mode(mode):
    """'packed' (default: one gradient row per Gaussian) or 'legacy' (atomics into the reference's six arrays); 'rows' exists in
    the ablate build of the library only and is refused by the product."""
    rc = _lib.hsr_set_backward_mode({"packed": 0, "rows": 1, "legacy": 2}[mode])
    if rc < 0:
        _fail(rc, "hsr_set_backward_mode")


def set_semantic_alpha(mode):
    """'reference' (default): the semantic loss never reaches alpha, as in the reference, whose backward stages the features for that
    term into an array nothing writes (RAST/cuda_rasterizer/backward.cu:778-779, :834-845).  'exact': the term those lines intend,
    as extra passes of the tile kernel (process-wide; needs the default packed accumulation mode).  Opt-in: not what the reference
    trains with."""
    rc = _lib.hsr_set_semantic_alpha_mode({"reference": 0, "exact": 1}[mode])
    if rc < 0:
        _fail(rc, "hsr_set_semantic_alpha_mode")


def get_semantic_alpha():
    return "exact" if int(_lib.hsr_get_semantic_alpha_mode()) == 1 else "reference"


def version():
    return _lib.hsr_version().decode()


def _fail(rc, what):
    raise RuntimeError("%s failed (code %d): %s" % (what, rc, _lib.hsr_last_error().decode()))


def _ptr(t):
    """Device pointer of a contiguous fp32/int32 tensor, or NULL for the reference's empty placeholders."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _prep(t, dev, dtype=torch.float32):
    """contiguous tensor on `dev` (reference: `.contiguous().data<float>()`, rasterize_points.cu:104-124)."""
    if t is None or t.numel() == 0:
        return None
    if t.device != dev:
        raise RuntimeError("diff_gaussian_rasterization: tensor on %s, expected %s" % (t.device, dev))
    if t.dtype != dtype:
        raise RuntimeError("diff_gaussian_rasterization: tensor dtype %s, expected %s" % (t.dtype, dtype))
    return t.contiguous()


def _require_gpu(means3D):
    if not means3D.is_cuda:
        raise RuntimeError("diff_gaussian_rasterization: tensors must live on a HIP device (got %s); "
                           "this build has no CPU path" % means3D.device)


_live_growers = {}          # key -> [tensor, device] of the hsr_buffers of forward calls in flight (keys ride in hsr_buffer.user)
_grower_keys = itertools.count(1)


def _grow_dispatch(nbytes, user):
    try:
        h = _live_growers[int(user)]
        h[0] = torch.empty(int(nbytes), dtype=torch.uint8, device=h[1])
        return h[0].data_ptr()
    except Exception:  # report failure through the C return code
        return None


# ONE callback object for the whole module: every ctypes callback is a reference cycle in itself (function pointer <-> thunk), so a
# callback per call that closes over its tensor kept the three state buffers of every forward alive until the cyclic collector ran
# (measured through this glue: one to two iterations' worth of device memory pending at any time)
_GROW_CB = _GROW_FN(_grow_dispatch)


class _Grower:
    """Adapts a torch uint8 tensor to hsr_buffer: pre-sized allocation + grow callback (the module's one callback, which finds this
    buffer through the key in hsr_buffer.user); close() when the library call has returned."""

    def __init__(self, nbytes, dev):
        self._key = next(_grower_keys)
        self._h = [torch.empty(int(nbytes), dtype=torch.uint8, device=dev), dev]
        _live_growers[self._key] = self._h
        self.buf = _HsrBuffer(self._h[0].data_ptr() if nbytes else None, int(nbytes), C.cast(_GROW_CB, C.c_void_p), self._key)

    @property
    def t(self):
        return self._h[0]

    def close(self):
        _live_growers.pop(self._key, None)


def _forward_common(semantic, background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier,
                    cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                    campos, prefiltered, debug, run_ahead=False):
    # run ahead of the device only where it has rendered this size before (the binning buffer is sized from that count)
    run_ahead = bool(run_ahead) and _async_forward and means3D.is_cuda and means3D.ndimension() == 2 and not debug
    key = None
    if run_ahead:
        key = (means3D.device.index, int(means3D.size(0)), int(image_width), int(image_height))
        run_ahead = key[1] > 0 and key in _binning_hint
        if run_ahead:
            _resolve_previous(key)
    if _ext is not None and means3D.is_cuda:
        stream = torch.cuda.current_stream(means3D.device).cuda_stream
        res = _ext.forward_common(bool(semantic), background, means3D, colors, semantics, opacity, scales, rotations,
                                  float(scale_modifier), cov3D_precomp, viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy),
                                  int(image_height), int(image_width), sh, int(degree), campos, bool(prefiltered), bool(debug),
                                  stream, run_ahead)
        if res[0] == HSR_PENDING:
            lz = _unresolved[key] = LazyRendered(res[10], key, stream)
            return (lz,) + tuple(res[1:10])
        return tuple(res[:10])
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:60-62
    _require_gpu(means3D)
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    K = 0
    if semantic:
        if semantics is not None and (semantics.numel() > 0 or semantics.ndimension() == 2):
            # the reference never checks this shape against its compile-time NUM_SEMANTIC (silent OOB)
            if semantics.ndimension() != 2 or semantics.size(0) != P:
                raise RuntimeError("semantics_precomp must have dimensions (num_points, K)")
            K = int(semantics.size(1))
    fopt = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        # every pixel of every output is written by the render kernel, so no zero-fill is needed
        # (the reference zero-fills with torch::full first, rasterize_points.cu:71-76)
        out_color = torch.empty((NUM_CHANNELS, H, W), **fopt)
        out_depth = torch.empty((1, H, W), **fopt)
        out_median = torch.empty((1, H, W), **fopt)
        out_opacity = torch.empty((1, H, W), **fopt)
        out_aux = torch.empty((K if semantic else 1, H, W), **fopt)  # semantic map or mask
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
        tens = [_prep(x, dev) for x in (background, means3D, sh, colors, semantics if semantic else None, opacity, scales,
                                        rotations, cov3D_precomp, viewmatrix, projmatrix, campos)]
        bg_, m3_, sh_, col_, sem_, op_, sc_, rot_, cov_, vm_, pm_, cp_ = tens
        if P == 0:
            geom = _Grower(0, dev); binning = _Grower(0, dev); img = _Grower(0, dev)
        else:
            geom = _Grower(_lib.hsr_required_geometry_bytes(P), dev)
            img = _Grower(_lib.hsr_required_image_bytes(W, H), dev)
            hint = _binning_hint.get((dev.index, P, W, H), 4 * P)
            # a call that runs ahead cannot grow the buffer afterwards: twice the last count instead of a quarter more
            binning = _Grower(_lib.hsr_required_binning_bytes(int(hint * (2.0 if run_ahead else 1.25)) + 1024), dev)
        ticket = None
        if run_ahead and P:
            ticket = _Ticket()
            _lib.hsr_forward_arm_async(C.byref(ticket))
        try:
            if semantic:
                rc = _lib.hsr_forward_semantic(
                    C.byref(geom.buf), C.byref(binning.buf), C.byref(img.buf), P, int(degree), M, K, _ptr(bg_), W, H, _ptr(m3_),
                    _ptr(sh_), _ptr(col_), _ptr(sem_), _ptr(op_), _ptr(sc_), float(scale_modifier), _ptr(rot_), _ptr(cov_),
                    _ptr(vm_), _ptr(pm_), _ptr(cp_), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                    _ptr(out_color), _ptr(out_aux), _ptr(out_depth), _ptr(out_median), _ptr(out_opacity), _ptr(radii),
                    int(bool(debug)), stream)
            else:
                rc = _lib.hsr_forward(
                    C.byref(geom.buf), C.byref(binning.buf), C.byref(img.buf), P, int(degree), M, _ptr(bg_), W, H, _ptr(m3_),
                    _ptr(sh_), _ptr(col_), _ptr(op_), _ptr(sc_), float(scale_modifier), _ptr(rot_), _ptr(cov_),
                    _ptr(vm_), _ptr(pm_), _ptr(cp_), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                    _ptr(out_color), _ptr(out_depth), _ptr(out_median), _ptr(out_opacity), _ptr(out_aux), _ptr(radii),
                    int(bool(debug)), stream)
        finally:
            for g_ in (geom, binning, img):
                g_.close()
        if rc == HSR_PENDING and ticket is not None:
            lz = _unresolved[key] = LazyRendered(ticket, key, stream)
            return lz, out_color, out_aux, out_depth, out_median, out_opacity, radii, geom.t, binning.t, img.t
        if rc < 0:
            _fail(rc, "rasterize_gaussians_semantic" if semantic else "rasterize_gaussians")
        if P:
            _binning_hint[(dev.index, P, W, H)] = rc
    return rc, out_color, out_aux, out_depth, out_median, out_opacity, radii, geom.t, binning.t, img.t


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, *, run_ahead=False):
    """RasterizeGaussiansCUDA (rasterize_points.cu:36-127) ->
    (rendered, color, depth, median_depth, opacity, mask, radii, geomBuffer, binningBuffer, imgBuffer)."""
    r, color, mask, depth, median, opac, radii, g, b, i = _forward_common(
        False, background, means3D, colors, None, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
        projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug, run_ahead)
    return r, color, depth, median, opac, mask, radii, g, b, i


def rasterize_gaussians_semantic(background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                                 degree, campos, prefiltered, debug, *, run_ahead=False):
    """RasterizeGaussiansCUDA_semantic (rasterize_points.cu:241-336) ->
    (rendered, color, semantic, depth, median_depth, opacity, radii, geomBuffer, binningBuffer, imgBuffer)."""
    r, color, sem, depth, median, opac, radii, g, b, i = _forward_common(
        True, background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
        projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug, run_ahead)
    return r, color, sem, depth, median, opac, radii, g, b, i


# Two per-call options beyond the reference's argument list (keyword-only, defaults = reference behaviour; the autograd
# node sets them from ctx.needs_input_grad — per call, so concurrent backward threads of several devices cannot see each
# other's choice):
#   want_cov3D_grad=False  cov3D_precomp needs no gradient (the usual case: Hier-SLAM passes scales and rotations):
#                          dL_dcov3D is neither allocated nor written and None is returned in its place.
#   geometry_only=True     no gradient is wanted for colours, opacities, semantics, scales, rotations, SH and cov3D (a
#                          tracking iteration: only the pose is optimised): the library forms the geometry sums only and the
#                          entry points return None for the rest.
def _backward_common(semantic, background, means3D, radii, colors, semantics, scales, rotations, scale_modifier,
                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_semantic,
                     dL_dout_depth, dL_dout_median_depth, dL_dout_final_opacity, sh, degree, campos, geomBuffer, R,
                     binningBuffer, imageBuffer, debug, want_cov3D_grad=True, geometry_only=False):
    R = int(R)   # a LazyRendered of a forward that ran ahead resolves here (and raises if its count did not fit the binning buffer)
    sunk = None
    if _gradient_sink is not None and means3D.is_cuda and int(means3D.size(0)) != 0:
        P_, K_ = int(means3D.size(0)), (int(dL_dout_semantic.size(0)) if semantic else 0)
        sunk = [_from_sink("raster." + n, shp, means3D.device) for n, shp in (
            ("means3D", (P_, 3)), ("colors_precomp", (P_, NUM_CHANNELS)), ("semantics_precomp", (P_, K_)), ("opacities", (P_, 1)),
            ("scales", (P_, 3)), ("rotations", (P_, 4)))]
    if _ext is not None and means3D.is_cuda:
        return _ext.backward_common(bool(semantic), background, means3D, radii, colors, semantics, scales, rotations,
                                    float(scale_modifier), cov3D_precomp, viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy),
                                    dL_dout_color, dL_dout_semantic, dL_dout_depth, dL_dout_median_depth, dL_dout_final_opacity,
                                    sh, int(degree), campos, geomBuffer, int(R), binningBuffer, imageBuffer, bool(debug),
                                    bool(want_cov3D_grad), bool(geometry_only),
                                    torch.cuda.current_stream(means3D.device).cuda_stream, sunk)
    _require_gpu(means3D)
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
    K = int(dL_dout_semantic.size(0)) if semantic else 0
    fopt = dict(dtype=torch.float32, device=dev)
    new = torch.zeros if P == 0 else torch.empty  # the library overwrites every element when P > 0
    packed_ok = P != 0 and int(_lib.hsr_backward_scratch_bytes(P, K, int(R))) > 0 and int(_lib.hsr_backward_scratch_bytes(P, K, 0)) > 0
    geo = bool(geometry_only) and packed_ok and colors is not None and colors.numel() != 0 and not _rows_or_legacy()
    pre = (lambda i: sunk[i] if (sunk is not None and sunk[i] is not None) else None)
    take = (lambda i, shape: pre(i) if pre(i) is not None else new(shape, **fopt))
    dL_dmeans3D = take(0, (P, 3))
    dL_dmeans2D = new((P, 3), **fopt)
    dL_dcolors = None if geo else take(1, (P, NUM_CHANNELS))
    dL_dsemantics = None if geo else take(2, (P, K))
    # with a scratch buffer (every mode but 'legacy') dL_dconic and dL_ddepths are intermediates nobody reads (the reference
    # keeps them inside RasterizeGaussiansBackwardCUDA, rasterize_points.cu:380-383): not allocated, not written;
    # dL_dcov3D only when the caller wants it (want_cov3D_grad=False from the autograd node when cov3D_precomp needs no grad)
    packed_scratch = P != 0 and int(_lib.hsr_backward_scratch_bytes(P, K, int(R))) > 0
    dL_dconic = None if packed_scratch else new((P, 2, 2), **fopt)
    dL_ddepths = None if packed_scratch else new((P, 1), **fopt)
    dL_dopacity = None if geo else take(3, (P, 1))
    dL_dcov3D = new((P, 6), **fopt) if ((want_cov3D_grad and not geo) or P == 0) else None
    dL_dsh = None if geo else new((P, M, 3), **fopt)
    dL_dscales = None if geo else take(4, (P, 3))
    dL_drotations = None if geo else take(5, (P, 4))
    if P != 0:
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            tens = [_prep(x, dev) for x in (background, means3D, sh, colors, semantics if semantic else None, scales,
                                            rotations, cov3D_precomp, viewmatrix, projmatrix, campos, dL_dout_color,
                                            dL_dout_semantic if semantic else None, dL_dout_depth, dL_dout_median_depth,
                                            dL_dout_final_opacity)]
            bg_, m3_, sh_, col_, sem_, sc_, rot_, cov_, vm_, pm_, cp_, gcol, gsem, gdep, gmed, gop = tens
            radii_ = _prep(radii, dev, torch.int32)
            # Backward scratch, sized by the library for the accumulation mode in force (include/hsr_rasterizer.h):
            # default: packed per-Gaussian gradient rows (P x stride floats) — halves the atomic requests
            nscratch = int(_lib.hsr_backward_scratch_bytes(P, K, int(R)))
            scratch = torch.empty(nscratch, dtype=torch.uint8, device=dev) if nscratch else None
            common_tail = (_ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drotations),
                           _ptr(scratch), nscratch, int(bool(debug)), stream)
            if semantic:
                rc = _lib.hsr_backward_semantic(
                    P, int(degree), M, K, int(R), _ptr(bg_), W, H, _ptr(m3_), _ptr(sh_), _ptr(col_), _ptr(sem_), _ptr(sc_),
                    float(scale_modifier), _ptr(rot_), _ptr(cov_), _ptr(vm_), _ptr(pm_), _ptr(cp_), float(tan_fovx),
                    float(tan_fovy), _ptr(radii_), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer),
                    _ptr(gcol), _ptr(gsem), _ptr(gdep), _ptr(gmed), _ptr(gop),
                    _ptr(dL_dmeans2D), _ptr(dL_dconic), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_dsemantics),
                    _ptr(dL_ddepths), *common_tail)
            else:
                rc = _lib.hsr_backward(
                    P, int(degree), M, int(R), _ptr(bg_), W, H, _ptr(m3_), _ptr(sh_), _ptr(col_), _ptr(sc_),
                    float(scale_modifier), _ptr(rot_), _ptr(cov_), _ptr(vm_), _ptr(pm_), _ptr(cp_), float(tan_fovx),
                    float(tan_fovy), _ptr(radii_), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer),
                    _ptr(gcol), _ptr(gdep), _ptr(gmed), _ptr(gop),
                    _ptr(dL_dmeans2D), _ptr(dL_dconic), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_ddepths), *common_tail)
            if rc < 0:
                _fail(rc, "rasterize_gaussians_backward_semantic" if semantic else "rasterize_gaussians_backward")
    return dL_dmeans2D, dL_dcolors, dL_dsemantics, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                 dL_dout_median_depth, dL_dout_final_opacity, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, debug, *, want_cov3D_grad=True, geometry_only=False):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:129-215) -> 8 tensors
    (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)."""
    g = _backward_common(False, background, means3D, radii, colors, None, scales, rotations, scale_modifier, cov3D_precomp,
                         viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, None, dL_dout_depth,
                         dL_dout_median_depth, dL_dout_final_opacity, sh, degree, campos, geomBuffer, R, binningBuffer,
                         imageBuffer, debug, want_cov3D_grad, geometry_only)
    return g[0], g[1], g[3], g[4], g[5], g[6], g[7], g[8]


def rasterize_gaussians_backward_semantic(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier,
                                          cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                          dL_dout_semantic, dL_dout_depth, dL_dout_median_depth, dL_dout_final_opacity, sh,
                                          degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug, *,
                                          want_cov3D_grad=True, geometry_only=False):
    """RasterizeGaussiansBackwardCUDA_semantic (rasterize_points.cu:340-432) -> 9 tensors
    (dL_dmeans2D, dL_dcolors, dL_dsemantics, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)."""
    return _backward_common(True, background, means3D, radii, colors, semantics, scales, rotations, scale_modifier,
                            cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_semantic,
                            dL_dout_depth, dL_dout_median_depth, dL_dout_final_opacity, sh, degree, campos, geomBuffer, R,
                            binningBuffer, imageBuffer, debug, want_cov3D_grad, geometry_only)


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (rasterize_points.cu:217-236) -> bool[P]."""
    _require_gpu(means3D)
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        with torch.cuda.device(dev):
            m3, vm, pm = (_prep(x, dev) for x in (means3D, viewmatrix, projmatrix))
            rc = _lib.hsr_mark_visible(P, _ptr(m3), _ptr(vm), _ptr(pm), present.data_ptr(),
                                       torch.cuda.current_stream(dev).cuda_stream)
            if rc < 0:
                _fail(rc, "mark_visible")
    return present


def state_layout(P, width, height, num_rendered):
    """Byte offsets of the fields inside the three opaque state buffers (parity tests / debugging)."""
    lay = _StateLayout()
    rc = _lib.hsr_get_state_layout(int(P), int(width), int(height), int(num_rendered), C.byref(lay))
    if rc < 0:
        _fail(rc, "hsr_get_state_layout")
    return {n: int(getattr(lay, n)) for n, _ in _StateLayout._fields_}
