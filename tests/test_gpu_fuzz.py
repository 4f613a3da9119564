"""Seeded random configurations through the whole rasterizer path against the oracle: ragged image sizes, any K, tiny and
huge splats, culled points, white / coloured backgrounds.  Same checks as test_gpu_parity._compare (integers bit-exact, floats
1e-4).  Sizes are small so that the oracle finishes in well under a second per case."""
import zlib

import numpy as np
import pytest

import scenes
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu


def _cases(n=28, seed=2024):
    g = np.random.default_rng(seed)
    out = []
    for i in range(n):
        W, H = int(g.integers(17, 200)), int(g.integers(17, 150))
        P = int(g.choice([1, 7, 63, 300, 1200, 2500]))
        semantic = bool(g.random() < 0.8)
        K = int(g.choice([0, 1, 3, 11, 16, 26, 27, 28, 31, 40, 52, 53, 74, 90])) if semantic else 0
        kind = str(g.choice(["slam", "aniso"]))
        sm = float(g.choice([0.3, 1.0, 3.0, 12.0, 60.0]))
        bg = tuple(float(x) for x in g.choice([0.0, 1.0, 0.3], size=3))
        behind = float(g.choice([0.0, 0.0, 0.3]))
        out.append(("%02d_%dx%d_P%d_K%d_%s_x%g" % (i, W, H, P, K, kind, sm), (W, H, P, K, kind, sm, semantic, "sr", bg, behind)))
    return out


CASES = _cases()


@pytest.mark.parametrize("name,cfg", CASES, ids=[c[0] for c in CASES])
def test_random_configuration(name, cfg):
    W, H, P, K, kind, sm, semantic, variant, bg, behind = cfg
    cam, sc, up = scenes.build(W, H, P, K, seed=zlib.crc32(name.encode()) % 1000, kind=kind, scale_mult=sm, bg=bg, behind_frac=behind)
    _compare(cam, sc, up, semantic, variant, None)


def test_threshold_ties_are_rare_and_bounded():
    """The blend accepts a splat iff alpha = min(0.99, o * exp(power)) >= 1/255 (reference forward.cu:492-496).  A splat whose alpha
    lands within an ulp of 1/255 can be accepted by one exp implementation and rejected by another (v_exp_f32 here, glibc in the
    oracle, CUDA's expf in the reference): the pixel then differs by that splat's whole contribution, at most (1/255) * T * |c|.
    Big anisotropic splats make such ties likelier; this pins the effect on two scenes found by a seed scan (tools/scan_seeds.py):
    a handful of pixels, each within one threshold contribution, everything else within the 1e-4 bar."""
    from harness import run_gpu, run_oracle
    W, H, P, K = 136, 141, 2500, 26
    for seed in (34, 40):
        cam, sc, up = scenes.build(W, H, P, K, seed=seed, kind="aniso", scale_mult=3.0, bg=(0, 0, 0), behind_frac=0.0)
        og, gg, sg = run_gpu(cam, sc, up, semantic=True, variant="sr")
        oo, go, so = run_oracle(cam, sc, up, semantic=True, variant="sr")
        assert np.array_equal(sg["keys"], so.field("keys")) and np.array_equal(og["radii"], oo["radii"])
        err = np.abs(np.asarray(og["color"], np.float64) - np.asarray(oo["color"], np.float64).reshape(3, H, W)).max(axis=0)
        bad = err > 1e-4
        assert 0 < int(bad.sum()) <= 8, int(bad.sum())          # the tie exists on these seeds, and it is a handful of pixels
        assert float(err.max()) <= 1.05 / 255.0                 # one splat at the 1/255 threshold, colour <= 1
