"""ctypes front-end to oracle/libhsr_oracle.so (the CPU restatement; TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libhsr_oracle.so")
_lib = None

FIELDS = dict(depths=0, means2D=1, conic_opacity=2, cov3D=3, rgb=4, clamped=5, radii=6, tiles_touched=7,
              point_offsets=8, keys_unsorted=9, keys=10, vals_unsorted=11, vals=12, ranges=13, final_T=14, n_contrib=15,
              median_pos=16, tie_pixels=17, tie_gaussians=18)


def build(force=False):
    src = os.path.join(_ROOT, "oracle", "hsr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-B", "libhsr_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.hsro_forward.restype = C.c_void_p
        _lib.hsro_field.restype = C.c_void_p
        _lib.hsro_field.argtypes = [C.c_void_p, C.c_int]
        _lib.hsro_num_rendered.argtypes = [C.c_void_p]
        _lib.hsro_free.argtypes = [C.c_void_p]
        _lib.hsro_get_higher_msb.restype = C.c_uint32
        _lib.hsro_get_higher_msb.argtypes = [C.c_uint32]
        _lib.hsro_set_median_rule.argtypes = [C.c_int]
        _lib.hsro_last_median_rule_disagreements.restype = C.c_long
    return _lib


def _f(a):
    if a is None:
        return None, C.c_void_p(0)
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if a.size == 0:
        return a, C.c_void_p(0)
    return a, a.ctypes.data_as(C.c_void_p)


def _np(x):
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


class OracleState:
    def __init__(self, handle, P, W, H, K, R, keep):
        self.h, self.P, self.W, self.H, self.K, self.R = handle, P, W, H, K, R
        self._keep = keep

    def field(self, name):
        L = lib()
        p = L.hsro_field(C.c_void_p(self.h), FIELDS[name])
        P, R, N = self.P, self.R, self.W * self.H
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        spec = dict(depths=(np.float32, (P,)), means2D=(np.float32, (P, 2)), conic_opacity=(np.float32, (P, 4)),
                    cov3D=(np.float32, (P, 6)), rgb=(np.float32, (P, 3)), clamped=(np.uint8, (P, 3)),
                    radii=(np.int32, (P,)), tiles_touched=(np.uint32, (P,)), point_offsets=(np.uint32, (P,)),
                    keys_unsorted=(np.uint64, (R,)), keys=(np.uint64, (R,)), vals_unsorted=(np.uint32, (R,)),
                    vals=(np.uint32, (R,)), ranges=(np.uint32, (T, 2)), final_T=(np.float32, (N,)),
                    n_contrib=(np.uint32, (N,)), median_pos=(np.uint32, (N,)), tie_pixels=(np.uint8, (N,)),
                    tie_gaussians=(np.uint8, (P,)))[name]
        n = int(np.prod(spec[1]))
        if n == 0:
            return np.zeros(spec[1], dtype=spec[0])
        buf = (C.c_char * (n * np.dtype(spec[0]).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=spec[0]).reshape(spec[1]).copy()

    def free(self):
        if self.h:
            lib().hsro_free(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def forward(cam, means3D, opacities, colors_precomp=None, shs=None, semantics_precomp=None, scales=None,
            rotations=None, cov3D_precomp=None, threads=0):
    """Oracle forward.  `cam` is a dict / NamedTuple with the GaussianRasterizationSettings fields.
    Returns (outputs dict, OracleState).  Semantic variant iff semantics_precomp is not None."""
    L = lib()
    if threads:
        L.hsro_set_threads(C.c_int(threads))
    g = (lambda k: cam[k]) if isinstance(cam, dict) else (lambda k: getattr(cam, k))
    W, H = int(g("image_width")), int(g("image_height"))
    means3D = _np(means3D)
    P = means3D.shape[0]
    shs_np = _np(shs)
    D = int(g("sh_degree"))
    M = 0 if shs_np is None or shs_np.size == 0 else shs_np.shape[1]
    sem_np = _np(semantics_precomp)
    semantic = sem_np is not None
    K = sem_np.shape[1] if semantic else 0
    N = W * H
    out_color = np.zeros((3, H, W), np.float32)
    out_sem = np.zeros((K, H, W), np.float32)
    out_depth = np.zeros((1, H, W), np.float32)
    out_median = np.zeros((1, H, W), np.float32)
    out_op = np.zeros((1, H, W), np.float32)
    out_mask = np.zeros((1, H, W), np.float32)
    radii = np.zeros((P,), np.int32)
    keep = []
    def f(a):
        arr, p = _f(_np(a))
        keep.append(arr)
        return p
    sem_ptr = f(sem_np)
    if semantic and K == 0:  # K==0 semantic: keep a non-null pointer so the semantic path is taken
        dummy = np.zeros(1, np.float32); keep.append(dummy); sem_ptr = dummy.ctypes.data_as(C.c_void_p)
    h = L.hsro_forward(C.c_int(P), C.c_int(D), C.c_int(M), C.c_int(K), f(g("bg")), C.c_int(W), C.c_int(H),
                       f(means3D), f(shs_np), f(colors_precomp), sem_ptr, f(opacities), f(scales),
                       C.c_float(float(g("scale_modifier"))), f(rotations), f(cov3D_precomp), f(g("viewmatrix")),
                       f(g("projmatrix")), f(g("campos")), C.c_float(float(g("tanfovx"))), C.c_float(float(g("tanfovy"))),
                       out_color.ctypes.data_as(C.c_void_p), out_sem.ctypes.data_as(C.c_void_p) if K else C.c_void_p(0),
                       out_depth.ctypes.data_as(C.c_void_p), out_median.ctypes.data_as(C.c_void_p),
                       out_op.ctypes.data_as(C.c_void_p), out_mask.ctypes.data_as(C.c_void_p),
                       radii.ctypes.data_as(C.c_void_p))
    if not h:
        raise MemoryError("hsro_forward failed")
    R = L.hsro_num_rendered(C.c_void_p(h))
    st = OracleState(h, P, W, H, K, R, keep)
    out = dict(num_rendered=R, color=out_color, depth=out_depth, median_depth=out_median, opacity=out_op, radii=radii)
    if semantic:
        out["semantic"] = out_sem
    else:
        out["mask"] = out_mask
    return out, st


def backward(st, cam, means3D, grads, colors_precomp=None, shs=None, semantics_precomp=None, scales=None,
             rotations=None, cov3D_precomp=None, threads=0, median_rule="reference"):
    """Oracle backward.  grads: dict(color[3,H,W], semantic[K,H,W]|None, depth, median, opacity).
    median_rule: "reference" = the splat the backward re-finds from its reconstructed T (backward.cu:623-626, :854-857);
    "forward" = the splat whose list position the forward recorded (what the HIP product does; identical except where the
    reconstructed T passes within rounding of 0.5).  The result carries `median_rule_disagreements`: the number of pixels on
    which the two rules pick differently."""
    L = lib()
    L.hsro_set_median_rule(C.c_int({"reference": 0, "forward": 1}[median_rule]))
    if threads:
        L.hsro_set_threads(C.c_int(threads))
    g = (lambda k: cam[k]) if isinstance(cam, dict) else (lambda k: getattr(cam, k))
    P, K = st.P, st.K
    shs_np = _np(shs)
    D = int(g("sh_degree"))
    M = 0 if shs_np is None or shs_np.size == 0 else shs_np.shape[1]
    keep = []
    def f(a):
        arr, p = _f(_np(a))
        keep.append(arr)
        return p
    o = dict(means2D=np.zeros((P, 3), np.float32), conic=np.zeros((P, 4), np.float32), opacities=np.zeros((P, 1), np.float32),
             colors_precomp=np.zeros((P, 3), np.float32), semantics_precomp=np.zeros((P, K), np.float32),
             depths=np.zeros((P, 1), np.float32), means3D=np.zeros((P, 3), np.float32), cov3D_precomp=np.zeros((P, 6), np.float32),
             shs=np.zeros((P, M, 3), np.float32), scales=np.zeros((P, 3), np.float32), rotations=np.zeros((P, 4), np.float32))
    def po(name):
        a = o[name]
        return a.ctypes.data_as(C.c_void_p) if a.size else C.c_void_p(0)
    has_scales = scales is not None and _np(scales).size > 0
    rc = L.hsro_backward(C.c_void_p(st.h), C.c_int(D), C.c_int(M), f(g("bg")), f(means3D), f(shs_np), f(colors_precomp),
                         f(semantics_precomp), f(scales), C.c_float(float(g("scale_modifier"))), f(rotations),
                         f(cov3D_precomp), f(g("viewmatrix")), f(g("projmatrix")), f(g("campos")),
                         C.c_float(float(g("tanfovx"))), C.c_float(float(g("tanfovy"))),
                         f(grads["color"]), f(grads.get("semantic")), f(grads["depth"]), f(grads["median"]), f(grads["opacity"]),
                         po("means2D"), po("conic"), po("opacities"), po("colors_precomp"), po("semantics_precomp"),
                         po("depths"), po("means3D"), po("cov3D_precomp"), po("shs"),
                         po("scales") if has_scales else C.c_void_p(0), po("rotations") if has_scales else C.c_void_p(0),
                         C.c_int(0))
    if rc != 0:
        raise RuntimeError("hsro_backward rc=%d" % rc)
    o["median_rule_disagreements"] = int(L.hsro_last_median_rule_disagreements())
    L.hsro_set_median_rule(C.c_int(0))
    return o


def get_higher_msb(n):
    return int(lib().hsro_get_higher_msb(C.c_uint32(n)))
