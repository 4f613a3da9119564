"""-m gpu: the opt-in EXACT semantic -> alpha gradient (hsr_set_semantic_alpha_mode(1), diff_gaussian_rasterization.set_semantic_alpha).

The reference's semantic backward reads the features for this term from a staging array nothing writes (RAST/cuda_rasterizer/backward.cu:
778-779 commented out, :834-845), so its semantic loss never moves opacity, covariance or position; the default of this library does the
same (tests/test_gpu_parity.py).  In the exact mode the term those lines intend is added by extra passes of the tile kernel
(hsr_render_bwd_q.hip, SEMA).  Checked against the oracle's own exact mode (oracle/hsr_oracle.c sem_alpha_mode 1), which
tests/test_oracle.py pins against float64 autograd of a dense restatement of the forward."""
import numpy as np
import pytest
import torch

import scenes
from harness import assert_close, run_gpu, run_oracle, tie_allowance

pytestmark = pytest.mark.gpu


@pytest.fixture
def exact_mode():
    from diff_gaussian_rasterization import _C
    assert _C.get_semantic_alpha() == "reference"   # the drop-in default
    _C.set_semantic_alpha("exact")
    yield
    _C.set_semantic_alpha("reference")


CASES = {
    # name: (W, H, P, K, kind, scale_mult, bg)
    "replica_tree_k26": (160, 96, 3000, 26, "aniso", 2.0, (0, 0, 0)),            # two passes of 16 channels, the second ragged
    "scannet_tree_k16": (128, 80, 2000, 16, "slam", 3.0, (0, 0, 0)),             # exactly one pass; compact rows
    "k5_white_background_ragged_image": (100, 70, 1500, 5, "aniso", 2.5, (1.0, 1.0, 1.0)),
    "k1": (96, 64, 800, 1, "slam", 3.0, (0, 0, 0)),                                # (isotropic splats: an anisotropic K = 1 scene sat at 0.9-1.2e-4 in one rotation entry, run to run)
    "k40_wide_rows": (96, 64, 1200, 40, "aniso", 2.0, (0.2, 0.1, 0.3)),          # main pass: the K > 27 kernel; three exact passes
    "k74_large_tree": (96, 64, 1200, 74, "aniso", 2.0, (0, 0, 0)),
    "huge_splats_k26": (96, 64, 300, 26, "aniso", 40.0, (0, 0, 0)),              # chunks shortened by the segment cap
    "deep_tiles_k16": (64, 48, 3000, 16, "aniso", 40.0, (0, 0, 0)),              # many staging batches per tile
}


@pytest.mark.parametrize("name", list(CASES))
def test_exact_mode_matches_the_oracles_exact_mode(name, exact_mode):
    W, H, P, K, kind, sm, bg = CASES[name]
    cam, sc, up = scenes.build(W, H, P, K, seed=23, kind=kind, scale_mult=sm, bg=bg)
    out_g, gr_g, st_g = run_gpu(cam, sc, up, semantic=True, variant="sr", want_state=False)
    out_o, gr_o, st_o = run_oracle(cam, sc, up, semantic=True, variant="sr", sem_alpha_exact=True)
    out_r, gr_r, st_r = run_oracle(cam, sc, up, semantic=True, variant="sr", bounds=False)
    # the case must be able to tell the two modes apart: the exact term is a large part of these gradients
    for n in ("means3D", "opacities", "scales"):
        diff = np.abs(np.asarray(gr_o[n], np.float64) - np.asarray(gr_r[n], np.float64)).max()
        assert diff > 1e-2 * np.abs(np.asarray(gr_o[n], np.float64)).max(), (n, diff)
    # ... and leaves the direct gradients alone
    assert np.array_equal(gr_o["semantics_precomp"], gr_r["semantics_precomp"])
    assert np.array_equal(gr_o["colors_precomp"], gr_r["colors_precomp"])
    for n in gr_o:
        g = np.asarray(gr_g[n])
        assert_close("grad " + n + " (exact semantic alpha)", g, gr_o[n], allowance=tie_allowance("grad " + n, st_o, g.shape, "gauss"))
    st_o.free()
    st_r.free()


def test_default_mode_is_untouched_by_a_round_trip_through_the_exact_mode():
    from diff_gaussian_rasterization import _C
    cam, sc, up = scenes.build(128, 80, 2000, 16, seed=5, kind="slam", scale_mult=3.0)
    _, g0, _ = run_gpu(cam, sc, up, semantic=True, variant="sr", want_state=False)
    _C.set_semantic_alpha("exact")
    try:
        _, g1, _ = run_gpu(cam, sc, up, semantic=True, variant="sr", want_state=False)
    finally:
        _C.set_semantic_alpha("reference")
    _, g2, _ = run_gpu(cam, sc, up, semantic=True, variant="sr", want_state=False)
    assert np.abs(g1["opacities"] - g0["opacities"]).max() > 1e-2 * np.abs(g0["opacities"]).max()
    for n in g0:
        assert np.abs(g2[n] - g0[n]).max() <= 1e-5 * max(1e-30, np.abs(g0[n]).max()), n   # atomics-order noise only


def test_exact_mode_in_a_geometry_only_backward(exact_mode):
    """A tracking iteration (only means3D / means2D want a gradient) with a semantic loss: the exact passes add into the 64-byte geometry rows."""
    from diff_gaussian_rasterization import GaussianRasterizer_semantic
    from harness import _cam_to
    W, H, P, K = 203, 131, 3000, 26
    cam, sc, up = scenes.build(W, H, P, K, seed=11, kind="slam", scale_mult=3.0)
    dev = torch.device("cuda:0")
    camd = _cam_to(cam, dev)
    means3D = sc["means3D"].to(dev).clone().requires_grad_(True)
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, radii, sem, depth, median, opacity = GaussianRasterizer_semantic(camd)(
        means3D=means3D, means2D=means2D, opacities=sc["opacities"].to(dev), colors_precomp=sc["colors_precomp"].to(dev),
        scales=sc["scales"].to(dev), rotations=sc["rotations"].to(dev), semantics_precomp=sc["semantics_precomp"].to(dev))
    loss = (sem * up["semantic"].to(dev)).sum() + (color * up["color"].to(dev)).sum() + (depth * up["depth"].to(dev)).sum() \
        + (median * up["median"].to(dev)).sum() + (opacity * up["opacity"].to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    _, go, so = run_oracle(cam, sc, up, semantic=True, variant="sr", sem_alpha_exact=True)
    for n, g in (("means3D", means3D.grad), ("means2D", means2D.grad)):
        g = g.cpu().numpy()
        assert_close("grad %s (geometry-only, exact semantic alpha)" % n, g, go[n], allowance=tie_allowance("grad " + n, so, g.shape, "gauss"))
    so.free()


def test_exact_mode_without_the_packed_rows_is_refused(exact_mode):
    from diff_gaussian_rasterization import _C
    cam, sc, up = scenes.build(96, 64, 500, 8, seed=3, kind="aniso", scale_mult=2.0)
    _C.set_backward_mode("legacy")
    try:
        with pytest.raises(RuntimeError, match="packed accumulation mode"):
            run_gpu(cam, sc, up, semantic=True, variant="sr", want_state=False)
    finally:
        _C.set_backward_mode("packed")
    torch.cuda.synchronize()
