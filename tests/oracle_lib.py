"""ctypes front-end to oracle/libhsr_oracle.so (the CPU restatement; TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libhsr_oracle.so")
_SO64 = os.path.join(_ROOT, "oracle", "libhsr_oracle_f64.so")   # the "truth" build: same fp32 lists, arithmetic in double
_SOFMA = os.path.join(_ROOT, "oracle", "libhsr_oracle_fma.so")  # the sensitivity build: fp32, FMA contraction allowed everywhere
_libs = {}

FIELDS = dict(depths=0, means2D=1, conic_opacity=2, cov3D=3, rgb=4, clamped=5, radii=6, tiles_touched=7,
              point_offsets=8, keys_unsorted=9, keys=10, vals_unsorted=11, vals=12, ranges=13, final_T=14, n_contrib=15,
              median_pos=16, tie_pixels=17, tie_gaussians=18, tie_img_bound=19)
IMG_BOUND_PLANES = ("color", "depth", "opacity", "semantic", "final_T", "mask")   # tie_img_bound[plane][pixel]


def build(force=False, precision="f32"):
    so = {"f64": _SO64, "fma": _SOFMA}.get(precision, _SO)
    srcs = [os.path.join(_ROOT, "oracle", n) for n in ("hsr_oracle.c", "hsr_oracle_la.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-B", os.path.basename(so)],
                              stdout=subprocess.DEVNULL)
    return so


class Bounds(C.Structure):
    """HsroBounds (oracle/hsr_oracle.c): tie bounds of the gradients, caller-allocated arrays of `real`"""
    _fields_ = [(n, C.c_void_p) for n in ("means2D", "opacities", "colors", "semantics", "means3D", "cov3D", "shs", "scales",
                                          "rotations")] + [("tie_pixels", C.c_long), ("tie_decisions", C.c_long),
                                                           ("overflow_pixels", C.c_long)]


def lib(precision="f32"):
    """precision "f32": the oracle; "f64": the truth build (images, final_T and gradients come back as float64); "fma": the
    contracted-FMA sensitivity build (oracle/Makefile) — never a parity checker"""
    L = _libs.get(precision)
    if L is None:
        L = C.CDLL(build(precision=precision))
        L.hsro_forward.restype = C.c_void_p
        L.hsro_field.restype = C.c_void_p
        L.hsro_field.argtypes = [C.c_void_p, C.c_int]
        L.hsro_num_rendered.argtypes = [C.c_void_p]
        L.hsro_free.argtypes = [C.c_void_p]
        L.hsro_get_higher_msb.restype = C.c_uint32
        L.hsro_get_higher_msb.argtypes = [C.c_uint32]
        L.hsro_set_median_rule.argtypes = [C.c_int]
        L.hsro_last_median_rule_disagreements.restype = C.c_long
        assert L.hsro_real_bytes() == (8 if precision == "f64" else 4)
        L.real = np.float64 if precision == "f64" else np.float32
        _libs[precision] = L
    return L


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
    def __init__(self, handle, P, W, H, K, R, keep, precision="f32"):
        self.h, self.P, self.W, self.H, self.K, self.R = handle, P, W, H, K, R
        self._keep = keep
        self.precision = precision

    def field(self, name):
        L = lib(self.precision)
        p = L.hsro_field(C.c_void_p(self.h), FIELDS[name])
        P, R, N = self.P, self.R, self.W * self.H
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        spec = dict(depths=(np.float32, (P,)), means2D=(np.float32, (P, 2)), conic_opacity=(np.float32, (P, 4)),
                    cov3D=(np.float32, (P, 6)), rgb=(np.float32, (P, 3)), clamped=(np.uint8, (P, 3)),
                    radii=(np.int32, (P,)), tiles_touched=(np.uint32, (P,)), point_offsets=(np.uint32, (P,)),
                    keys_unsorted=(np.uint64, (R,)), keys=(np.uint64, (R,)), vals_unsorted=(np.uint32, (R,)),
                    vals=(np.uint32, (R,)), ranges=(np.uint32, (T, 2)), final_T=(L.real, (N,)),
                    n_contrib=(np.uint32, (N,)), median_pos=(np.uint32, (N,)), tie_pixels=(np.uint8, (N,)),
                    tie_gaussians=(np.uint8, (P,)), tie_img_bound=(L.real, (len(IMG_BOUND_PLANES), N)))[name]
        n = int(np.prod(spec[1]))
        if n == 0:
            return np.zeros(spec[1], dtype=spec[0])
        buf = (C.c_char * (n * np.dtype(spec[0]).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=spec[0]).reshape(spec[1]).copy()

    def img_bound(self, name):
        """how far the output `name` of each pixel moves when its flagged threshold decisions go the other way ([N]; 0 elsewhere)"""
        return self.field("tie_img_bound")[IMG_BOUND_PLANES.index(name)]

    def free(self):
        if self.h:
            lib(self.precision).hsro_free(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def forward(cam, means3D, opacities, colors_precomp=None, shs=None, semantics_precomp=None, scales=None,
            rotations=None, cov3D_precomp=None, threads=0, precision="f32"):
    """Oracle forward.  `cam` is a dict / NamedTuple with the GaussianRasterizationSettings fields.
    Returns (outputs dict, OracleState).  Semantic variant iff semantics_precomp is not None.
    precision "f64": the truth build (float64 images; identical integer state)."""
    L = lib(precision)
    rt = L.real
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
    out_color = np.zeros((3, H, W), rt)
    out_sem = np.zeros((K, H, W), rt)
    out_depth = np.zeros((1, H, W), rt)
    out_median = np.zeros((1, H, W), rt)
    out_op = np.zeros((1, H, W), rt)
    out_mask = np.zeros((1, H, W), rt)
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
    st = OracleState(h, P, W, H, K, R, keep, precision)
    out = dict(num_rendered=R, color=out_color, depth=out_depth, median_depth=out_median, opacity=out_op, radii=radii)
    if semantic:
        out["semantic"] = out_sem
    else:
        out["mask"] = out_mask
    return out, st


# HSR_TEST_SEM_ALPHA=exact: every oracle backward that does not say otherwise runs the exact semantic -> alpha mode, and tests/conftest.py
# switches the library to it — the parity and fuzz tests then exercise the opt-in mode end to end (tools/r04_exact_fuzz.sh)
DEFAULT_SEM_ALPHA_EXACT = os.environ.get("HSR_TEST_SEM_ALPHA", "") == "exact"


def backward(st, cam, means3D, grads, colors_precomp=None, shs=None, semantics_precomp=None, scales=None,
             rotations=None, cov3D_precomp=None, threads=0, median_rule="reference", bounds=False, fp32_atomics_seed=None, exp_ulps=0.0,
             sem_alpha_exact=None, arg_roundings=0.0):
    """Oracle backward.  grads: dict(color[3,H,W], semantic[K,H,W]|None, depth, median, opacity).
    median_rule: "reference" = the splat the backward re-finds from its reconstructed T (backward.cu:623-626, :854-857);
    "forward" = the splat whose list position the forward recorded (what the HIP product does; identical except where the
    reconstructed T passes within rounding of 0.5).  The result carries `median_rule_disagreements`: the number of pixels on
    which the two rules pick differently.
    sem_alpha_exact: the semantic loss also reaches alpha (oracle sem_alpha_mode 1; the product's opt-in mode, default off = reference as observed).
    bounds=True: also the tie bounds of the gradients (oracle/hsr_oracle.c, "Threshold ties") as o["bounds"][name] — how far
    each gradient entry moves when the flagged threshold decisions of the flagged pixels are taken the other way."""
    if sem_alpha_exact is None:
        sem_alpha_exact = DEFAULT_SEM_ALPHA_EXACT
    L = lib(st.precision)
    rt = L.real
    # fp32_atomics_seed: the per-Gaussian sums are accumulated in fp32 in a seeded random tile order — one of the orders the reference's
    # (or the HIP kernels') fp32 atomicAdds can arrive in — instead of in double (oracle/hsr_oracle.c hsro_set_accumulation)
    # exp_ulps (fp32 model only): the exponential of every (pixel, splat) pair is off by up to that many ulps, hashed from the seed —
    # a GPU's exp (v_exp_f32: 1 ulp; CUDA's expf: 2 ulp documented) instead of glibc's
    L.hsro_set_accumulation(C.c_int(0 if fp32_atomics_seed is None else 1), C.c_uint(int(fp32_atomics_seed or 0)))
    L.hsro_set_exp_error(C.c_float(float(exp_ulps) if fp32_atomics_seed is not None else 0.0))
    # arg_roundings (fp32 model only): ... and the exponent's argument rounded as another correct fp32 evaluation order would (hsro_set_exp_argument_error)
    L.hsro_set_exp_argument_error(C.c_float(float(arg_roundings) if fp32_atomics_seed is not None else 0.0))
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
    o = dict(means2D=np.zeros((P, 3), rt), conic=np.zeros((P, 4), rt), opacities=np.zeros((P, 1), rt),
             colors_precomp=np.zeros((P, 3), rt), semantics_precomp=np.zeros((P, K), rt),
             depths=np.zeros((P, 1), rt), means3D=np.zeros((P, 3), rt), cov3D_precomp=np.zeros((P, 6), rt),
             shs=np.zeros((P, M, 3), rt), scales=np.zeros((P, 3), rt), rotations=np.zeros((P, 4), rt))
    def po(name):
        a = o[name]
        return a.ctypes.data_as(C.c_void_p) if a.size else C.c_void_p(0)
    has_scales = scales is not None and _np(scales).size > 0
    bstruct, barr = None, None
    if bounds:
        barr = dict(means2D=np.zeros((P, 3), rt), opacities=np.zeros((P, 1), rt), colors_precomp=np.zeros((P, 3), rt),
                    semantics_precomp=np.zeros((P, K), rt), means3D=np.zeros((P, 3), rt), cov3D_precomp=np.zeros((P, 6), rt),
                    shs=np.zeros((P, M, 3), rt), scales=np.zeros((P, 3), rt), rotations=np.zeros((P, 4), rt))
        bstruct = Bounds()
        for cname, pname in (("means2D", "means2D"), ("opacities", "opacities"), ("colors", "colors_precomp"),
                             ("semantics", "semantics_precomp"), ("means3D", "means3D"), ("cov3D", "cov3D_precomp"), ("shs", "shs"),
                             ("scales", "scales"), ("rotations", "rotations")):
            setattr(bstruct, cname, barr[pname].ctypes.data if barr[pname].size else None)
    rc = L.hsro_backward(C.c_void_p(st.h), C.c_int(D), C.c_int(M), f(g("bg")), f(means3D), f(shs_np), f(colors_precomp),
                         f(semantics_precomp), f(scales), C.c_float(float(g("scale_modifier"))), f(rotations),
                         f(cov3D_precomp), f(g("viewmatrix")), f(g("projmatrix")), f(g("campos")),
                         C.c_float(float(g("tanfovx"))), C.c_float(float(g("tanfovy"))),
                         f(grads["color"]), f(grads.get("semantic")), f(grads["depth"]), f(grads["median"]), f(grads["opacity"]),
                         po("means2D"), po("conic"), po("opacities"), po("colors_precomp"), po("semantics_precomp"),
                         po("depths"), po("means3D"), po("cov3D_precomp"), po("shs"),
                         po("scales") if has_scales else C.c_void_p(0), po("rotations") if has_scales else C.c_void_p(0),
                         C.c_int(1 if sem_alpha_exact else 0), C.byref(bstruct) if bstruct is not None else C.c_void_p(0))
    if rc != 0:
        raise RuntimeError("hsro_backward rc=%d" % rc)
    if bounds:
        o["bounds"] = barr
        o["bounds_info"] = dict(tie_pixels=int(bstruct.tie_pixels), tie_decisions=int(bstruct.tie_decisions),
                                overflow_pixels=int(bstruct.overflow_pixels))
    o["median_rule_disagreements"] = int(L.hsro_last_median_rule_disagreements())
    L.hsro_set_median_rule(C.c_int(0))
    L.hsro_set_accumulation(C.c_int(0), C.c_uint(0))
    L.hsro_set_exp_error(C.c_float(0.0))
    L.hsro_set_exp_argument_error(C.c_float(0.0))
    return o


def get_higher_msb(n):
    return int(lib().hsro_get_higher_msb(C.c_uint32(n)))
