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

FUZZ_OUTLIERS = [(9, "537_88x79_P2500_K53_aniso_x3"), (9, "565_170x112_P2500_K74_aniso_x3"), (9, "798_69x39_P300_K1_aniso_x1"),
                 (5, "106_175x18_P2500_K53_aniso_x3"), (5, "677_168x126_P2500_K3_aniso_x1")]
REPORTS = {}


def _scene(name, seed, n=1000):
    cfg = dict(_cases(n, seed))[name]
    W, H, P, K, kind, sm, semantic, variant, bg, behind = cfg
    cam, sc, up = scenes.build(W, H, P, K, seed=zlib.crc32(name.encode()) % 1000, kind=kind, scale_mult=sm, bg=bg, behind_frac=behind)
    return cam, sc, up, semantic, variant


def _check(name, rep):
    REPORTS[name] = rep
    assert rep["lists_equal"]
    for tname, t in rep["tensors"].items():
        h, o = t["hip_vs_truth"], t["oracle32_vs_truth"]
        # (1) against the truth HIP meets the bar the north star sets against the reference: 1e-4, tensor-wide and element-wise
        assert h["err_over_max"] <= 1e-4, (name, tname, h)
        # (2) element-wise it may exceed 1e-4 only where the fp32 ORACLE is itself that far from the truth (conditioning): HIP is
        #     never more than 3x farther from the truth than the oracle is, plus the rounding floor
        assert h["elementwise"] <= max(1e-4, 3.0 * o["elementwise"] + 2e-5), (name, tname, h, o)


@pytest.mark.parametrize("seed,name", FUZZ_OUTLIERS, ids=[n for _, n in FUZZ_OUTLIERS])
def test_fuzz_outliers_against_the_truth(seed, name):
    cam, sc, up, semantic, variant = _scene(name, seed)
    _check(name, truth_report(cam, sc, up, semantic=semantic, variant=variant))


@pytest.mark.parametrize("cfg", [(136, 141, 2500, 26, "aniso", 3.0, 34), (136, 141, 2500, 8, "aniso", 3.0, 455), (320, 200, 20000, 26, "slam", 1.0, 3),
                                 (96, 64, 1200, 74, "aniso", 2.0, 11)])
def test_hip_and_oracle_are_equidistant_from_the_truth(cfg):
    W, H, P, K, kind, sm, seed = cfg
    cam, sc, up = scenes.build(W, H, P, K, seed=seed, kind=kind, scale_mult=sm)
    _check("%dx%d_P%d_K%d_%s_x%g_s%d" % cfg, truth_report(cam, sc, up, semantic=True))


def teardown_module(module):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "truth_report.json"), "w") as fh:
            json.dump(REPORTS, fh, indent=1, sort_keys=True)
    except Exception:
        pass
