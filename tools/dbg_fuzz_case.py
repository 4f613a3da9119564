"""One case of the fuzz generator (tests/test_gpu_fuzz.py) taken apart: how far HIP, the fp32 oracle and the fp32 models of the noise floor
sit from the truth build, per gradient.  Usage (GPU box): HSR_FUZZ_CASES=3000 HSR_FUZZ_SEED=4242 python tools/dbg_fuzz_case.py 2821_185x33"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "hier-slam_amd"), os.path.join(ROOT, "tests")]
import harness  # noqa: E402
import oracle_lib as O  # noqa: E402
import scenes  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402

pat = sys.argv[1]
name, cfg = [c for c in F.CASES if pat in c[0]][0]
W, H, P, K, kind, sm, semantic, variant, bg, behind = cfg
cam, sc, up = scenes.build(W, H, P, K, seed=zlib.crc32(name.encode()) % 1000, kind=kind, scale_mult=sm, bg=bg, behind_frac=behind)
print(name)
runs = [harness.run_gpu(cam, sc, up, semantic=semantic, variant=variant, want_state=False)[1] for _ in range(3)]
_, gr_o, st_o = harness.run_oracle(cam, sc, up, semantic=semantic, variant=variant)
_, gr_t, st_t = harness.run_oracle(cam, sc, up, semantic=semantic, variant=variant, precision="f64", bounds=False)
kw = harness.variant_kwargs(sc, variant, None)
if semantic:
    kw["semantics_precomp"] = sc["semantics_precomp"]
g_ = {n: v.numpy() for n, v in up.items()}
if not semantic:
    g_["semantic"] = None
models = {}
for label, e, a in (("orders + exp 1 ulp", 1.0, 0.0), ("orders + exp 1 ulp + argument 1.5 roundings", 1.0, 1.5), ("orders + argument 0.5", 0.0, 0.5)):
    models[label] = [O.backward(st_o, cam, sc["means3D"], g_, median_rule="forward", fp32_atomics_seed=s, exp_ulps=e, arg_roundings=a, **kw) for s in range(8)]
for n in ("means3D", "scales", "rotations", "opacities"):
    t = np.asarray(gr_t[n], np.float64)
    mx = np.abs(t).max()
    fl = harness.floor_for("grad " + n)
    def dist(a):
        d = np.abs(np.asarray(a, np.float64).reshape(t.shape) - t)
        e = d / np.maximum(np.abs(t), fl * mx)
        return d.max() / mx, e.max(), np.unravel_index(e.argmax(), e.shape)
    print("== grad", n, "max", mx, "floor frac", fl)
    for i, r in enumerate(runs):
        print("  HIP run %d: tensor-wide %.3e element-wise %.3e at %s" % ((i,) + dist(r[n])))
    print("  fp32 oracle: tensor-wide %.3e element-wise %.3e at %s" % dist(gr_o[n]))
    for label, ms in models.items():
        ds = [dist(m[n]) for m in ms]
        print("  %s: tensor-wide max %.3e element-wise max %.3e (each: %s)" % (label, max(d[0] for d in ds), max(d[1] for d in ds), " ".join("%.2e" % d[1] for d in ds)))
    w = dist(runs[0][n])[2]
    row = w[0]
    print("  worst HIP row %d: truth %s\n      HIP %s\n      o32 %s" % (row, t[row], np.asarray(runs[0][n])[row], np.asarray(gr_o[n])[row]))
    print("      scales %s opacity %s" % (sc["scales"][row].numpy(), sc["opacities"][row].numpy()))
