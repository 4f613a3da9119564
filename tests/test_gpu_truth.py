"""-m gpu: HIP and the fp32 oracle, each against the TRUTH build of the oracle (oracle/libhsr_oracle_f64.so: the same fp32
preprocess, keys and lists; compositing, its backward and the per-Gaussian chain rule in double).

Why.  Every other parity test compares HIP with the fp32 oracle.  Where the two differ by more than the element-wise bound
there are three explanations — a defect in a kernel, a threshold decision taken the other way (covered by the oracle's tie
bounds), or two correct fp32 evaluations of an ill-conditioned expression landing apart — and only a third, more exact
evaluation can tell the last from the first: if HIP and the oracle are EQUIDISTANT from the truth, neither is wrong; if the
oracle sits on the truth and HIP does not, HIP is.  The named cases are the three the 1 000-case fuzz run of round 2 left
unexplained (HSR_FUZZ_CASES=1000 HSR_FUZZ_SEED=9: one `grad means3D` element each at 1.6e-4 element-wise vs the fp32 oracle) and the
two that round 3's first run of the same generator with seed 5 produced (a `grad means3D` element at 1.05e-4, a `grad rotations`
element at 1.26e-4).  Which cases land outside the bound changes from run to run (fp32 atomics order); the class does not."""
import json
import os
import zlib

import numpy as np
import pytest

import scenes
from harness import truth_report
from test_gpu_fuzz import _cases

pytestmark = pytest.mark.gpu

# (fuzz seed, case name[, number of cases generated]) — every case a longer fuzz run ever flagged
FUZZ_OUTLIERS = [(9, "537_88x79_P2500_K53_aniso_x3"), (9, "565_170x112_P2500_K74_aniso_x3"), (9, "798_69x39_P300_K1_aniso_x1"),
                 (5, "106_175x18_P2500_K53_aniso_x3"), (5, "677_168x126_P2500_K3_aniso_x1"),
                 # late round 3, 3 000 cases: dL_drotations of one 20:1 needle that covers all 56 tiles: HIP 1.4e-4 of max from the truth in
                 # every run, the fp32 oracle 0.3e-4, eight fp32 arrival orders of the reference's formulation up to 1.0e-4
                 (4242, "2571_101x121_P2500_K16_aniso_x3", 3000)]
REPORTS = {}


def _scene(name, seed, n=1000):
    cfg = dict(_cases(n, seed))[name]
    W, H, P, K, kind, sm, semantic, variant, bg, behind = cfg
    cam, sc, up = scenes.build(W, H, P, K, seed=zlib.crc32(name.encode()) % 1000, kind=kind, scale_mult=sm, bg=bg, behind_frac=behind)
    return cam, sc, up, semantic, variant


SEEDS = (0, 1, 2, 3, 4, 5, 6, 7)   # fp32-atomics model: eight arrival orders of the per-Gaussian sums (tests/harness.truth_report)


def _check(name, rep):
    """HIP meets the 1e-4 bar against the truth — tensor-wide and element-wise — except where fp32 arithmetic itself cannot: there
    HIP may be at most twice as far from the truth as the farther of (the fp32 oracle, the fp32-atomics model of the reference's own
    accumulation over eight arrival orders), plus the rounding floor.  A defect in a kernel shows as HIP alone being far."""
    REPORTS[name] = rep
    assert rep["lists_equal"]
    for tname, t in rep["tensors"].items():
        h, o = t["hip_vs_truth"], t["oracle32_vs_truth"]
        m = t.get("fp32_atomics_model_vs_truth", o)
        for key in ("err_over_max", "elementwise"):
            floor = max(o[key], m[key])
            assert h[key] <= max(1e-4, 2.0 * floor + 2e-5), (name, tname, key, h, o, m)


@pytest.mark.parametrize("case", FUZZ_OUTLIERS, ids=[c[1] for c in FUZZ_OUTLIERS])
def test_fuzz_outliers_against_the_truth(case):
    seed, name = case[0], case[1]
    cam, sc, up, semantic, variant = _scene(name, seed, *case[2:])
    _check(name, truth_report(cam, sc, up, semantic=semantic, variant=variant, atomics_seeds=SEEDS))


@pytest.mark.parametrize("cfg", [(136, 141, 2500, 26, "aniso", 3.0, 34), (136, 141, 2500, 8, "aniso", 3.0, 455), (320, 200, 20000, 26, "slam", 1.0, 3),
                                 (96, 64, 1200, 74, "aniso", 2.0, 11)])
def test_hip_and_oracle_are_equidistant_from_the_truth(cfg):
    W, H, P, K, kind, sm, seed = cfg
    cam, sc, up = scenes.build(W, H, P, K, seed=seed, kind=kind, scale_mult=sm)
    _check("%dx%d_P%d_K%d_%s_x%g_s%d" % cfg, truth_report(cam, sc, up, semantic=True, atomics_seeds=SEEDS if P <= 5000 else ()))


# The comparisons of the default suite with the least margin (tests/conftest.py names them at the end of every run: "worst grad means3D /
# scales / rotations").  As of round 4: dL_dmeans3D element-wise — fuzz case 05 (63 big anisotropic splats on a 29 x 121
# image); dL_dscales — fuzz case 21 (2 500 elongated splats, 136 x 141); dL_drotations — case v2_14 of the variants generator (scale modifier
# 1.7).  Their HIP-vs-truth and oracle-vs-truth distances go on record in gpurun_out/truth_report.json with the other cases.
WORST_MARGIN = [("v1", "05_29x121_P63_K26_aniso_x12"), ("v1", "21_136x141_P2500_K28_aniso_x3"), ("v1", "12_127x18_P1_K16_slam_x60"),
                ("v2", "v2_14_89x140_P2500_K11_aniso_x1_sr_rgb_mod1.7_ctr")]


@pytest.mark.parametrize("case", WORST_MARGIN, ids=[c[1] for c in WORST_MARGIN])
def test_worst_margin_cases_of_the_suite(case):
    gen, name = case
    if gen == "v1":
        cam, sc, up, semantic, variant = _scene(name, 2024, 28)
        extra = None
    else:
        from test_gpu_fuzz import CASES_V2, build_v2
        cam, sc, up, semantic, variant, extra = build_v2(name, dict(CASES_V2)[name])
    _check("worst-margin " + name, truth_report(cam, sc, up, semantic=semantic, variant=variant, extra=extra, atomics_seeds=SEEDS))


def teardown_module(module):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "truth_report.json"), "w") as fh:
            json.dump(REPORTS, fh, indent=1, sort_keys=True)
    except Exception:
        pass
