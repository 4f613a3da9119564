"""-m gpu: the HIP path (through the public torch API and the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): radii, tiles_touched, offsets, sorted keys, values and tile ranges
BIT-EXACT; images and gradients within 1e-4 (fp32).  n_contrib / median depth depend on hard thresholds
fed by exp(), whose last-ulp differs between glibc and the GPU: compared with a small mismatch budget."""
import os

import numpy as np
import pytest
import torch

import scenes
from harness import assert_close, run_gpu, run_oracle

pytestmark = pytest.mark.gpu

CASES = {
    # name: (W, H, P, K, kind, scale_mult, semantic, variant, bg, behind_frac)
    "replica_tree_k26": (160, 96, 3000, 26, "aniso", 2.0, True, "sr", (0, 0, 0), 0.0),
    "scannet_tree_k16": (128, 80, 2000, 16, "slam", 3.0, True, "sr", (0, 0, 0), 0.0),
    "generic_k5_white_bg": (100, 70, 1500, 5, "aniso", 2.5, True, "sr", (1.0, 1.0, 1.0), 0.0),  # ragged W,H + bg term
    "generic_k40_two_chunks": (96, 64, 1200, 40, "aniso", 2.0, True, "sr", (0, 0, 0), 0.0),
    "large_tree_k74": (96, 64, 1200, 74, "aniso", 2.0, True, "sr", (0.2, 0.1, 0.3), 0.0),
    "flat_k102": (80, 48, 800, 102, "slam", 3.0, True, "sr", (0, 0, 0), 0.0),
    "odd_k33": (96, 64, 1200, 33, "aniso", 2.0, True, "sr", (0, 0, 0), 0.0),     # unaligned rows, 2 channel blocks
    "odd_k75_ragged": (100, 70, 1500, 75, "aniso", 2.5, True, "sr", (0.3, 0.2, 0.1), 0.0),
    "k52_four_column_groups": (100, 70, 1500, 52, "aniso", 2.5, True, "sr", (0.1, 0.2, 0.3), 0.0),   # bf16-split contraction, 4 groups
    "k124_widest_single_pass": (64, 48, 600, 124, "slam", 3.0, True, "sr", (0, 0, 0), 0.0),
    "k130_chunked": (64, 48, 600, 130, "slam", 3.0, True, "sr", (0, 0, 0), 0.0),
    "k257_many_chunks": (48, 40, 300, 257, "aniso", 3.0, True, "sr", (0.1, 0.0, 0.2), 0.0),   # 9 forward chunks, 4 backward passes, odd K
    "wide_deep_tiles_k76": (64, 48, 1500, 76, "aniso", 40.0, True, "sr", (0, 0, 0), 0.0),  # many batches, early termination
    "plain_mask": (144, 96, 2500, 0, "aniso", 2.0, False, "sr", (0, 0, 0), 0.0),
    "plain_cov3d": (96, 80, 1500, 0, "aniso", 2.0, False, "cov", (0, 0, 0), 0.0),
    "culled_behind_camera": (96, 64, 1500, 26, "aniso", 2.0, True, "sr", (0, 0, 0), 0.4),
    "huge_splats": (96, 64, 300, 26, "aniso", 40.0, True, "sr", (0, 0, 0), 0.0),  # splats covering the whole screen
    "semantic_k0": (64, 48, 500, 0, "aniso", 2.0, True, "sr", (0, 0, 0), 0.0),
    # > 2048 entries in every tile: the per-tile sort's block-radix fallback (and many staging batches per tile)
    "deep_tiles_3000": (64, 48, 3000, 16, "aniso", 40.0, True, "sr", (0, 0, 0), 0.0),
    "deep_tiles_20000": (64, 48, 20000, 3, "aniso", 40.0, True, "sr", (0, 0, 0), 0.0),   # 20 000 entries in every tile: the in-tile radix sort, 80+ batches
    # BASELINE.json configs[0]: 256x256, 5k Gaussians, RGB + depth (the plain renderer), at its real size
    "config0_256x256_5k_plain": (256, 256, 5000, 0, "slam", 1.0, False, "sr", (0, 0, 0), 0.0),
    # BASELINE.json configs[3]'s frame: ScanNet 640x480, NYU40 tree K = 16 (one rank's keyframe of the 8-GPU batch)
    "config3_scannet_640x480_k16": (640, 480, 60000, 16, "slam", 1.0, True, "sr", (0, 0, 0), 0.0),
}


TIE_BOUNDED = []   # comparisons that needed the oracle's tie bound (reported at the end of the run: tests/conftest.py)
CONDITIONED = []   # comparisons settled against the truth build: HIP within twice the fp32 noise floor of that gradient (ditto)


def _compare(cam, sc, up, semantic, variant, extra=None, grad_rtol=1e-4):
    out_g, gr_g, st_g = run_gpu(cam, sc, up, semantic=semantic, variant=variant, extra=extra)
    out_o, gr_o, st_o = run_oracle(cam, sc, up, semantic=semantic, variant=variant, extra=extra)
    # ---- integer outputs: bit-exact ----
    assert np.array_equal(out_g["radii"], out_o["radii"]), "radii"
    assert st_g["num_rendered"] == out_o["num_rendered"], "num_rendered"
    assert np.array_equal(st_g["tiles_touched"], st_o.field("tiles_touched")), "tiles_touched"
    assert np.array_equal(st_g["point_offsets"], st_o.field("point_offsets")), "point_offsets"
    assert np.array_equal(st_g["keys"], st_o.field("keys")), "sorted keys"
    assert np.array_equal(st_g["vals"], st_o.field("vals")), "sorted values"
    assert np.array_equal(st_g["ranges"], st_o.field("ranges")), "tile ranges"
    vis = out_o["radii"] > 0
    # per-Gaussian fp32 state feeding integer outputs: identical bits by construction (no contraction)
    assert np.array_equal(st_g["depths"][vis], st_o.field("depths")[vis]), "depths"
    assert np.array_equal(st_g["means2D"][vis], st_o.field("means2D")[vis]), "means2D"
    assert np.array_equal(st_g["conic_opacity"][vis], st_o.field("conic_opacity")[vis]), "conic_opacity"
    # ---- thresholded integers: budgeted ----
    npix = out_o["color"].shape[1] * out_o["color"].shape[2]
    nmis = int((st_g["n_contrib"] != st_o.field("n_contrib")).sum())
    assert nmis <= max(2, npix // 2000), "n_contrib mismatches: %d" % nmis
    mmis = int((st_g["median_pos"] != st_o.field("median_pos")).sum())   # the recorded T = 0.5 crossing: same kind of tie
    assert mmis <= max(2, npix // 2000), "median_pos mismatches: %d" % mmis
    # ---- images and gradients: strict first.  Where that fails, the oracle says which pixels took a decision within a few ulps of its
    # threshold (alpha >= 1/255, T(1 - alpha) < 1e-4, power > 0, T crossing 0.5) — there v_exp_f32 and glibc's expf may decide
    # differently and the pixel differs by a whole contribution — and HOW FAR each output / gradient entry moves when such a decision
    # goes the other way (oracle/hsr_oracle.c "Threshold ties": every flagged decision evaluated both ways).  The comparison is then
    # repeated with TIE_SLACK x that bound added entry by entry: nothing is left out, an error larger than one flipped decision
    # explains still fails, and the rows whose allowance exceeds their ordinary bound are counted and capped. ----
    tie_pix = st_o.field("tie_pixels").astype(bool).reshape(-1)
    assert int(tie_pix.sum()) <= max(8, npix // 200), "tie-risk pixels: %d of %d" % (int(tie_pix.sum()), npix)
    assert st_o.bounds_info["overflow_pixels"] == 0, st_o.bounds_info

    truth = {}   # the truth build's gradients and the fp32 noise floor per tensor, computed at most once per comparison

    def conditioned(name, got, allowance):
        """Third tier, gradients only: HIP and the fp32 oracle are two fp32 evaluations; where they differ by more than the bound the
        TRUTH build (same lists, arithmetic in double) says who is right.  The noise floor of a gradient = how far the fp32 oracle and
        eight runs of the fp32 model of another correct implementation (the reference's own fp32-atomics accumulation in seeded arrival orders, a 1-ulp
        exp, the exponent's argument rounded as another evaluation order would: harness.FP32_MODEL_*), and the fp32 oracle with one upstream gradient at a
        time (harness.fp32_error_samples), sit from the truth; HIP passes if
        it is within twice that floor (+ rounding), tensor-wide and element-wise — a defect shows as HIP alone being far.  (Eight orders, not
        four: the largest of a handful of draws from a heavy-tailed spread is a shaky floor — case 2571 of the 3 000-case run with seed
        4242, a 20:1 needle covering all 56 tiles, sat at 1.4e-4 against a four-order floor of 0.5e-4 and an eight-order floor of 1.0e-4.)"""
        import harness
        import oracle_lib as O
        if not truth:
            out_t, gr_t, st_t = run_oracle(cam, sc, up, semantic=semantic, variant=variant, extra=extra, precision="f64", bounds=False)
            truth["s"] = harness.fp32_error_samples(st_o, st_t, cam, sc, up, semantic, variant, extra)
            st_t.free()
            kw_ = harness.variant_kwargs(sc, variant, extra)
            if semantic:
                kw_["semantics_precomp"] = sc["semantics_precomp"]
            g_ = {n: (v.numpy() if hasattr(v, "numpy") else v) for n, v in up.items()}
            if not semantic:
                g_["semantic"] = None
            truth["t"] = gr_t
            truth["m"] = [O.backward(st_o, cam, sc["means3D"], g_, median_rule="forward", fp32_atomics_seed=seed, exp_ulps=harness.FP32_MODEL_EXP_ULPS,
                                     arg_roundings=harness.FP32_MODEL_ARG_ROUNDINGS, **kw_) for seed in range(8)]
        key = name.replace("grad ", "")
        t = np.asarray(truth["t"][key], np.float64).reshape(np.asarray(got).shape)
        mx = max(float(np.abs(t).max()), 1e-30)
        fl = harness.floor_for(name)

        def dist(a):
            d = np.maximum(np.abs(np.asarray(a, np.float64).reshape(t.shape) - t) - allowance, 0.0)
            return float(d.max() / mx), float((d / np.maximum(np.abs(t), fl * mx)).max())
        h = dist(got)
        def dist_pair(a, b):   # the fp32 oracle against the truth with one upstream gradient alone (harness.fp32_error_samples)
            # an ABSOLUTE error of one term mix of the same sum, measured on the scale HIP's distance is measured on — the full gradient's
            # maximum and floor (normalised by the single-upstream run's own, smaller scale it inflated the floor)
            b = np.asarray(b, np.float64).reshape(t.shape)
            d = np.abs(np.asarray(a, np.float64).reshape(t.shape) - b)
            return float(d.max() / mx), float((d / np.maximum(np.abs(t), fl * mx)).max())
        floor = [max(x) for x in zip(dist(gr_o[key]), *[dist(m[key]) for m in truth["m"]], *[dist_pair(go_[key], gt_[key]) for go_, gt_ in truth["s"]])]
        for hv, fv, what in zip(h, floor, ("tensor-wide", "element-wise")):
            assert hv <= max(1e-4, 2.0 * fv + 2e-5), "%s: %s distance from the truth %.3e, fp32 noise floor %.3e" % (name, what, hv, fv)
        CONDITIONED.append((name, h[1], floor[1]))

    def close(name, got, exp, per, **kw):
        try:
            assert_close(name, got, exp, **kw)
        except AssertionError:
            import harness
            allowance = harness.tie_allowance(name, st_o, np.asarray(got).shape, per)
            try:
                if not (allowance > 0).any():
                    raise
                harness.OBSERVED.pop()   # the strict attempt's record: replaced by the bounded one below
                assert_close(name + " (beyond the oracle's tie bound)", got, exp, allowance=allowance, **kw)
                if per == "gauss":
                    loose = harness.loosened_rows(np.asarray(exp, np.float64).reshape(np.asarray(got).shape), allowance, harness.floor_for(name))
                    rows = np.asarray(got).shape[0]
                    assert loose <= max(harness.TIE_LOOSENED_MIN, int(harness.TIE_LOOSENED_FRAC * rows)), \
                        "%s: %d of %d rows are loosened by the tie bound" % (name, loose, rows)
                TIE_BOUNDED.append(name)
            except AssertionError:
                if per != "gauss" or np.asarray(got).shape[0] > 20000:    # images are never settled this way; nor full-size scenes (single-thread model)
                    raise
                conditioned(name, got, allowance)

    names = ["color", "depth", "opacity"] + (["semantic"] if semantic else ["mask"])
    for n in names:
        close(n, out_g[n], out_o[n], "pixel")
    med_bad = int((np.abs(out_g["median_depth"] - out_o["median_depth"]) > 1e-4).sum())
    assert med_bad <= max(2, npix // 2000), "median depth outliers: %d" % med_bad
    close("final_T", st_g["final_T"], st_o.field("final_T"), "pixel")
    # ---- gradients ----
    for n in gr_o:
        assert n in gr_g, n
        close("grad " + n, gr_g[n], gr_o[n], "gauss", rtol=grad_rtol, atol=1e-4)
    st_o.free()


@pytest.mark.parametrize("name", list(CASES))
def test_parity(name):
    W, H, P, K, kind, sm, semantic, variant, bg, behind = CASES[name]
    cam, sc, up = scenes.build(W, H, P, K, seed=11, kind=kind, scale_mult=sm, bg=bg, behind_frac=behind)
    extra = {"cov3D_precomp": scenes.cov3d_from_scene(sc)} if variant == "cov" else None
    _compare(cam, sc, up, semantic, variant, extra)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_parity_sh(deg):
    W, H, P, K = 96, 64, 1000, 16
    cam, sc, up = scenes.build(W, H, P, K, seed=5, kind="aniso", scale_mult=2.0)
    cam["sh_degree"] = deg
    extra = {"shs": scenes.random_sh(P, 16)}
    _compare(cam, sc, up, True, "sr", extra)


def test_no_gaussians():
    from diff_gaussian_rasterization import GaussianRasterizer_semantic
    from harness import _cam_to
    cam, sc, up = scenes.build(64, 48, 0, 26)
    dev = torch.device("cuda:0")
    z = lambda *s: torch.zeros(*s, device=dev)
    outs = GaussianRasterizer_semantic(_cam_to(cam, dev))(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1),
                                                          colors_precomp=z(0, 3), scales=z(0, 3), rotations=z(0, 4),
                                                          semantics_precomp=z(0, 26))
    color, radii, sem, depth, median, opac = outs
    # reference: P == 0 leaves the zero-filled outputs untouched (rasterize_points.cu:294-295)
    assert radii.numel() == 0 and float(color.abs().max()) == 0 and float(median.abs().max()) == 0
    assert sem.shape == (26, 48, 64) and float(sem.abs().max()) == 0


def test_all_culled():
    """every Gaussian behind the camera: num_rendered = 0, empty ranges, median depth stays at its default 15"""
    cam, sc, up = scenes.build(64, 48, 200, 16, behind_frac=1.0)
    sc["means3D"][:, 2] = -torch.abs(sc["means3D"][:, 2]) - 1.0  # behind the (untilted-ish) camera for sure
    out_g, gr_g, st_g = run_gpu(cam, sc, up, semantic=True)
    out_o, gr_o, st_o = run_oracle(cam, sc, up, semantic=True)
    if out_o["num_rendered"] == 0:
        assert st_g["num_rendered"] == 0
        assert float(np.abs(out_g["median_depth"] - 15.0).max()) == 0
        assert float(np.abs(out_g["color"]).max()) == 0
    for n in gr_o:
        assert_close("grad " + n, gr_g[n], gr_o[n])


def test_mark_visible():
    from diff_gaussian_rasterization import GaussianRasterizer
    from harness import _cam_to
    import oracle_lib as O
    import ctypes as C
    cam, sc, up = scenes.build(64, 48, 3000, 0, behind_frac=0.5)
    dev = torch.device("cuda:0")
    vis = GaussianRasterizer(_cam_to(cam, dev)).markVisible(sc["means3D"].to(dev)).cpu().numpy()
    P = sc["means3D"].shape[0]
    exp = np.zeros(P, np.uint8)
    m = np.ascontiguousarray(sc["means3D"].numpy()); v = np.ascontiguousarray(cam["viewmatrix"].numpy()); p = np.ascontiguousarray(cam["projmatrix"].numpy())
    O.lib().hsro_mark_visible(C.c_int(P), m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p),
                              exp.ctypes.data_as(C.c_void_p))
    assert np.array_equal(vis, exp.astype(bool))
    assert 0 < vis.sum() < P


def test_repeatable_and_saved_state_independent():
    """two forwards before two backwards (legal in torch): each backward must see its own state buffers"""
    from diff_gaussian_rasterization import GaussianRasterizer_semantic
    from harness import _cam_to
    cam, sc, up = scenes.build(96, 64, 1500, 16, seed=3)
    cam2, sc2, up2 = scenes.build(96, 64, 1500, 16, seed=4)
    dev = torch.device("cuda:0")
    def fwd(cam_, sc_):
        leaf = {n: sc_[n].to(dev).clone().requires_grad_(True) for n in ("means3D", "opacities", "colors_precomp", "scales", "rotations", "semantics_precomp")}
        outs = GaussianRasterizer_semantic(_cam_to(cam_, dev))(means2D=torch.zeros(1500, 3, device=dev, requires_grad=True), **leaf)
        return leaf, outs
    l1, o1 = fwd(cam, sc)
    l2, o2 = fwd(cam2, sc2)
    (o1[0] * up["color"].to(dev)).sum().backward()
    (o2[0] * up2["color"].to(dev)).sum().backward()
    up_only_color = {n: (v if n == "color" else torch.zeros_like(v)) for n, v in up.items()}
    _, g1, _ = run_oracle(cam, sc, up_only_color)
    assert_close("grad means3D (first of two in flight)", l1["means3D"].grad.cpu().numpy(), g1["means3D"])


def test_speculative_forward_and_its_fallback_agree_with_the_oracle(monkeypatch):
    """The forward enqueues emit / per-tile sort / render before it has read num_rendered back, with array bases resolved on the
    device (BinDevRef); if the caller's binning buffer is too small those kernels return at once and the call grows the buffer and
    runs them again.  Both ways, and with speculation switched off, the state and outputs must match the oracle bit for bit /
    to 1e-4 (_compare), and num_rendered must be the same."""
    import torch
    from diff_gaussian_rasterization import _C
    if os.environ.get("HSR_ASYNC_FORWARD"):
        pytest.skip("the blocking forward's grow-and-rerun fallback: with the non-blocking forward forced on, a hint of 1 is the overflow that "
                    "mode reports by design (test_non_blocking_forward_runs_ahead_and_fails_loudly_when_the_buffer_was_too_small)")
    W, H, P, K = 203, 131, 3000, 26
    cam, sc, up = scenes.build(W, H, P, K, seed=5, kind="slam")
    key = (torch.device("cuda:0").index, P, W, H)
    _C._binning_hint.pop(key, None)
    _compare(cam, sc, up, True, "sr", None)                 # default hint 4P: roomy, speculative path
    R = _C._binning_hint[key]
    assert R > 0
    monkeypatch.setitem(_C._binning_hint, key, 1)           # far too small: fallback path (the call grows the buffer)
    _compare(cam, sc, up, True, "sr", None)
    assert _C._binning_hint[key] == R
    monkeypatch.setitem(_C._binning_hint, key, int(R / 1.25) - 900)   # required bytes just below what R needs
    _compare(cam, sc, up, True, "sr", None)
    assert _C._binning_hint[key] == R


@pytest.mark.parametrize("semantic,K", [(True, 26), (True, 74), (False, 0)])
def test_geometry_only_backward_matches_the_full_one(semantic, K):
    """When only means3D / means2D want a gradient (a tracking iteration optimises the camera pose alone) the library forms the
    geometry sums only (hsr_backward*: dL_dcolor, dL_dopacity, dL_dsemantics NULL -> render_bwd_geo_kernel).  dL_dmeans3D and
    dL_dmeans2D must equal those of a full backward of the same render (to atomics-order noise) and the oracle's (1e-4)."""
    import torch
    from diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizer_semantic, _C
    from harness import _cam_to, run_oracle, assert_close
    W, H, P = 203, 131, 3000
    cam, sc, up = scenes.build(W, H, P, K, seed=11, kind="slam", scale_mult=3.0)
    dev = torch.device("cuda:0")
    camd = _cam_to(cam, dev)
    grads = {}
    for mode in ("pose", "full"):
        means3D = sc["means3D"].to(dev).clone().requires_grad_(True)
        means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
        rest = {n: sc[n].to(dev).clone().requires_grad_(mode == "full") for n in ("opacities", "colors_precomp", "scales", "rotations")}
        seen = {}
        real = _C._lib.hsr_backward_semantic if semantic else _C._lib.hsr_backward
        if semantic:
            sem = sc["semantics_precomp"].to(dev).clone().requires_grad_(mode == "full")
            outs = GaussianRasterizer_semantic(camd)(means3D=means3D, means2D=means2D, opacities=rest["opacities"], colors_precomp=rest["colors_precomp"],
                                                     scales=rest["scales"], rotations=rest["rotations"], semantics_precomp=sem)
            color, radii, semantic_map, depth, median, opacity = outs
            loss = (semantic_map * up["semantic"].to(dev)).sum()
        else:
            outs = GaussianRasterizer(camd)(means3D=means3D, means2D=means2D, opacities=rest["opacities"], colors_precomp=rest["colors_precomp"],
                                            scales=rest["scales"], rotations=rest["rotations"])
            color, radii, depth, median, opacity, mask = outs
            loss = 0.0
        loss = loss + (color * up["color"].to(dev)).sum() + (depth * up["depth"].to(dev)).sum() + (median * up["median"].to(dev)).sum() \
            + (opacity * up["opacity"].to(dev)).sum()
        loss.backward()
        torch.cuda.synchronize()
        grads[mode] = (means3D.grad.cpu().numpy(), means2D.grad.cpu().numpy())
        if mode == "pose":
            assert rest["colors_precomp"].grad is None and rest["opacities"].grad is None
    for a, b, name in ((grads["pose"][0], grads["full"][0], "means3D"), (grads["pose"][1], grads["full"][1], "means2D")):
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(b).max()), name
    oo, go, so = run_oracle(cam, sc, up, semantic=semantic, variant="sr")
    assert_close("means3D (geometry-only) vs oracle", grads["pose"][0], go["means3D"])
    assert_close("means2D (geometry-only) vs oracle", grads["pose"][1], go["means2D"])


def test_backward_argument_checks_in_a_ctypes_child():
    import subprocess
    import sys
    code = r'''
import sys; sys.path[:0]=['hier-slam_amd','tests']
import torch, scenes
from diff_gaussian_rasterization import GaussianRasterizer_semantic, _C
from harness import _cam_to
cam, sc, up = scenes.build(96, 64, 500, 5, seed=1, kind="slam")
dev = torch.device("cuda:0"); camd = _cam_to(cam, dev)
def run(pose_only):
    m3 = sc["means3D"].to(dev).clone().requires_grad_(True); m2 = torch.zeros(500, 3, device=dev, requires_grad=True)
    rest = {n: sc[n].to(dev).clone().requires_grad_(not pose_only) for n in ("opacities", "colors_precomp", "scales", "rotations", "semantics_precomp")}
    outs = GaussianRasterizer_semantic(camd)(means3D=m3, means2D=m2, opacities=rest["opacities"], colors_precomp=rest["colors_precomp"],
                                             scales=rest["scales"], rotations=rest["rotations"], semantics_precomp=rest["semantics_precomp"])
    (outs[0].sum() + outs[3].sum()).backward(); torch.cuda.synchronize(); return m3.grad.clone()
_C.set_backward_mode("legacy")
a = run(True)            # geometry-only request in the legacy mode: the glue must fall back to a full backward, not fail
b = run(False)
assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
# straight at the C ABI: legacy mode (no scratch) with dL_dconic = NULL is refused
rc = _C._lib.hsr_backward_semantic(500, 0, 0, 5, 1, None, 96, 64, None, None, None, None, None, 1.0, None, None, None, None, None, 1.0, 1.0,
                                   None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, 0, 0, None)
assert rc < 0, rc
print("ok")
'''
    env = dict(os.environ, HSR_GLUE="ctypes")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _render_sem(cam, sc, dev, **flags):
    from diff_gaussian_rasterization import GaussianRasterizer_semantic
    from harness import _cam_to
    c = dict(cam)
    c.update(flags)
    leaf = {n: sc[n].to(dev).clone().requires_grad_(True) for n in ("means3D", "opacities", "colors_precomp", "scales", "rotations",
                                                                    "semantics_precomp")}
    m2 = torch.zeros(sc["means3D"].shape[0], 3, device=dev, requires_grad=True)
    return leaf, GaussianRasterizer_semantic(_cam_to(c, dev))(means2D=m2, **leaf)


def test_prefiltered_violation_is_an_error_not_a_device_trap():
    """prefiltered=True promises that every point passes the frustum test; the reference __trap()s the device when one does
    not (auxiliary.h:156-160).  Here the call raises with the reference's message and the context keeps working."""
    dev = torch.device("cuda:0")
    cam, sc, up = scenes.build(96, 64, 1500, 16, seed=2, behind_frac=0.3)     # 30 % of the points behind / too near the camera
    with pytest.raises(RuntimeError, match="filtered although prefiltered is set"):
        _render_sem(cam, sc, dev, prefiltered=True)
    torch.cuda.synchronize()
    # the same scene without the promise renders, and so does a promise that holds — through the whole comparison
    _compare(cam, sc, up, True, "sr", None)
    cam2, sc2, up2 = scenes.build(96, 64, 1500, 16, seed=2, behind_frac=0.0)
    cam2["prefiltered"] = True
    _compare(cam2, sc2, up2, True, "sr", None)


def test_debug_mode_checks_every_launch_and_dumps_snapshots(tmp_path, monkeypatch):
    """raster_settings.debug=True (auxiliary.h:166-173 sync-and-throw after every launch; __init__.py:82-90 / :294-301
    input snapshots on failure): parity with every launch checked, snapshot_fw.dump / snapshot_bw.dump written on an
    injected failure and loadable without unpickling code."""
    from diff_gaussian_rasterization import _C
    monkeypatch.chdir(tmp_path)
    dev = torch.device("cuda:0")
    cam, sc, up = scenes.build(96, 64, 1200, 16, seed=4)
    cam["debug"] = True
    _compare(cam, sc, up, True, "sr", None)               # per-launch checks on, no speculation: same results
    assert not os.path.exists("snapshot_fw.dump") and not os.path.exists("snapshot_bw.dump")
    bad = dict(sc)
    bad["semantics_precomp"] = sc["semantics_precomp"][:-7]          # wrong row count: the forward refuses it
    with pytest.raises(RuntimeError, match="semantics_precomp"):
        _render_sem(cam, bad, dev)
    snap = torch.load("snapshot_fw.dump", weights_only=True)
    assert isinstance(snap, tuple) and snap[1].shape == (1200, 3) and snap[3].shape == (1193, 16) and not snap[1].is_cuda
    # backward: inject a failure below the autograd node
    leaf, outs = _render_sem(cam, sc, dev)
    def boom(*a, **k):
        raise RuntimeError("injected backward failure")
    monkeypatch.setattr(_C, "rasterize_gaussians_backward_semantic", boom)
    with pytest.raises(RuntimeError, match="injected backward failure"):
        outs[0].sum().backward()
    snap = torch.load("snapshot_bw.dump", weights_only=True)
    assert isinstance(snap, tuple) and snap[1].shape == (1200, 3) and int(snap[-4]) == outs[0].grad_fn.num_rendered
    # without debug nothing is dumped
    os.remove("snapshot_fw.dump"); os.remove("snapshot_bw.dump")
    cam["debug"] = False
    leaf, outs = _render_sem(cam, sc, dev)
    with pytest.raises(RuntimeError):
        outs[0].sum().backward()
    assert not os.path.exists("snapshot_bw.dump")


def _fwd_bwd(cam, sc, up, dev):
    leaf, outs = _render_sem(cam, sc, dev)
    color, radii, sem, depth, median, opac = outs
    upd = {n: v.to(dev) for n, v in up.items()}
    loss = (color * upd["color"]).sum() + (sem * upd["semantic"]).sum() + (depth * upd["depth"]).sum() \
        + (median * upd["median"]).sum() + (opac * upd["opacity"]).sum()
    loss.backward()
    return [o.detach() for o in (color, sem, depth, median, opac)], radii, {n: t.grad for n, t in leaf.items()}


def _same_up_to_atomics_order(ref, got):
    """two runs of the same computation whose float atomics arrived in a different order: the sums themselves (colours, semantics,
    opacities, means2D) agree to a few 1e-6 of their maximum; what passes through the conic -> covariance -> scale / rotation chain is
    amplified — on the anisotropic test scenes two runs of ONE scene differ by up to 2e-4 of the maximum in dL_drotations (measured:
    1.9e-5 typical, 2.1e-4 once in five runs) — so those are held to 1e-3: this is a check for races and bleed-through, whose signature
    is a wrong image or a gradient that is off by its own size, not a parity check (that is _compare)."""
    for n in ref:
        tol = 1e-3 if n in ("means3D", "scales", "rotations") else 2e-5
        assert float((ref[n] - got[n]).abs().max()) <= tol * max(1.0, float(ref[n].abs().max())), n


def test_two_renders_on_two_non_default_streams_match_the_default_stream():
    """The C ABI takes the launch stream explicitly and the glue passes torch's CURRENT stream; the forward's speculative tail and
    the host-mapped num_rendered slot (one per thread and device, hsr_api.hip) must not assume the default stream.  Two scenes are
    rendered forward-forward-backward-backward on two side streams, interleaved, and compared with their default-stream runs:
    images and radii bit for bit (the forward has no atomics), gradients to atomics-order noise."""
    dev = torch.device("cuda:0")
    scenes_ = [scenes.build(203, 131, 3000, 26, seed=5, kind="slam"), scenes.build(160, 96, 2500, 16, seed=6, kind="aniso", scale_mult=2.0)]
    ref = [_fwd_bwd(cam, sc, up, dev) for cam, sc, up in scenes_]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    pend = []
    for (cam, sc, up), st in zip(scenes_, streams):
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            leaf, outs = _render_sem(cam, sc, dev)
            pend.append((leaf, outs, {n: v.to(dev) for n, v in up.items()}))
    got = []
    for (leaf, outs, upd), st in zip(pend, streams):
        with torch.cuda.stream(st):
            color, radii, sem, depth, median, opac = outs
            ((color * upd["color"]).sum() + (sem * upd["semantic"]).sum() + (depth * upd["depth"]).sum()
             + (median * upd["median"]).sum() + (opac * upd["opacity"]).sum()).backward()
            got.append(([o.detach() for o in (color, sem, depth, median, opac)], radii, {n: t.grad for n, t in leaf.items()}))
    for st in streams:
        st.synchronize()
    for (imgs_r, radii_r, gr_r), (imgs_g, radii_g, gr_g) in zip(ref, got):
        assert torch.equal(radii_r, radii_g)
        for a, b in zip(imgs_r, imgs_g):
            assert torch.equal(a, b)
        _same_up_to_atomics_order(gr_r, gr_g)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs a second GPU")
def test_second_device_matches_the_first():
    """the same render on cuda:1 (device guard, per-device read-back slot, per-device binning hint)"""
    cam, sc, up = scenes.build(203, 131, 3000, 26, seed=5, kind="slam")
    a = _fwd_bwd(cam, sc, up, torch.device("cuda:0"))
    b = _fwd_bwd(cam, sc, up, torch.device("cuda:1"))
    assert torch.equal(a[1].cpu(), b[1].cpu())
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x.cpu(), y.cpu())
    _same_up_to_atomics_order({n: v.cpu() for n, v in a[2].items()}, {n: v.cpu() for n, v in b[2].items()})


@pytest.mark.parametrize("glue", ["compiled", "ctypes"])
def test_gradient_exchange_bucket_is_written_by_the_backward_itself(glue):
    """hsr_utils.parallel.GradientExchange on one rank: after backward() every leaf's .grad IS a slice of the exchange bucket (the
    backward wrote it there through the gradient sink, autograd adopted it) and holds the same values as an ordinary run."""
    import subprocess
    import sys
    code = r'''
import sys; sys.path[:0]=['hier-slam_amd','tests']
import torch, scenes
from diff_gaussian_rasterization import GaussianRasterizer_semantic
from hsr_utils.parallel import GradientExchange
from harness import _cam_to
cam, sc, up = scenes.build(160, 96, 2500, 26, seed=3, kind="aniso", scale_mult=2.0)
dev = torch.device("cuda:0"); camd = _cam_to(cam, dev)
names = ("means3D", "colors_precomp", "semantics_precomp", "opacities", "scales", "rotations")
def run(ex):
    leaf = {n: sc[n].to(dev).clone().requires_grad_(True) for n in names}
    if ex == "x":
        ex = GradientExchange({"raster." + n: leaf[n] for n in names}, dev, depth=2)
    outs = GaussianRasterizer_semantic(camd)(means2D=torch.zeros(2500, 3, device=dev, requires_grad=True), **leaf)
    if ex is not None:
        ex.begin_step(outs[1])
    torch.autograd.backward([outs[0], outs[2], outs[3], outs[4], outs[5]],
                            [up[n].to(dev) for n in ("color", "semantic", "depth", "median", "opacity")])
    if ex is not None:
        for i, n in enumerate(names):
            assert leaf[n].grad.data_ptr() == ex.buckets[0].views[i].data_ptr(), n
        ex.submit(); ex.drain()
        st = ex.stats(); assert st["zero_copy_tensors"] == 6 and st["copied_tensors"] == 0, st
    torch.cuda.synchronize()
    return {n: leaf[n].grad.clone() for n in names}
a, b = run(None), run("x")
for n in names:
    assert float((a[n] - b[n]).abs().max()) <= 1e-5 * max(1.0, float(a[n].abs().max())), n
print("ok")
'''
    env = dict(os.environ)
    if glue == "ctypes":
        env["HSR_GLUE"] = "ctypes"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_non_blocking_forward_runs_ahead_and_fails_loudly_when_the_buffer_was_too_small():
    """diff_gaussian_rasterization.set_async_forward(True): a forward that a backward will follow returns before the device has
    counted num_rendered (hsr_forward_arm_async); the count is a LazyRendered the backward resolves.  Same images bit for bit and
    the same gradients as the blocking call; without gradients, and on the first call of a size, the call stays blocking; and if
    num_rendered does not fit the binning buffer the outputs are NaN and resolving the count raises — then the next call fits."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    W, H, P, K = 203, 131, 3000, 26
    cam, sc, up = scenes.build(W, H, P, K, seed=5, kind="slam")
    key = (dev.index, P, W, H)
    ref = _fwd_bwd(cam, sc, up, dev)                  # blocking; leaves the binning hint of this size
    prev = dgr.set_async_forward(True)
    try:
        assert key in _C._binning_hint
        leaf, outs = _render_sem(cam, sc, dev)
        node = outs[0].grad_fn
        assert isinstance(node.num_rendered, _C.LazyRendered), type(node.num_rendered)
        upd = {n: v.to(dev) for n, v in up.items()}
        color, radii, sem, depth, median, opac = outs
        ((color * upd["color"]).sum() + (sem * upd["semantic"]).sum() + (depth * upd["depth"]).sum()
         + (median * upd["median"]).sum() + (opac * upd["opacity"]).sum()).backward()
        torch.cuda.synchronize()
        assert int(node.num_rendered) == _C._binning_hint[key] > 0
        for a, b in zip(ref[0], (color, sem, depth, median, opac)):
            assert torch.equal(a, b.detach())
        assert torch.equal(ref[1], radii)
        for n in ref[2]:
            assert float((ref[2][n] - leaf[n].grad).abs().max()) <= 1e-5 * max(1.0, float(ref[2][n].abs().max())), n
        # no gradient wanted: nobody would resolve the count, so the call stays blocking
        with torch.no_grad():
            _, outs2 = _render_sem(cam, {n: v for n, v in sc.items()}, dev)
        leaf3 = {n: sc[n].to(dev) for n in ("means3D", "opacities", "colors_precomp", "scales", "rotations", "semantics_precomp")}
        from diff_gaussian_rasterization import GaussianRasterizer_semantic
        from harness import _cam_to
        o3 = GaussianRasterizer_semantic(_cam_to(cam, dev))(means2D=torch.zeros(P, 3, device=dev), **leaf3)
        assert o3[0].grad_fn is None and torch.equal(o3[0], ref[0][0])
        # overflow: a hint far too small -> the call runs ahead into a buffer that cannot hold num_rendered
        R = _C._binning_hint[key]
        _C._binning_hint[key] = 8
        leaf, outs = _render_sem(cam, sc, dev)
        torch.cuda.synchronize()
        assert bool(torch.isnan(outs[0]).all()) and bool(torch.isnan(outs[2]).all()) and bool(torch.isnan(outs[5]).all())
        with pytest.raises(RuntimeError, match="does not fit the binning buffer"):
            outs[0].sum().backward()
        assert _C._binning_hint[key] == R                                   # the count that did not fit sizes the next buffer
        leaf, outs = _render_sem(cam, sc, dev)
        assert isinstance(outs[0].grad_fn.num_rendered, _C.LazyRendered)
        outs[0].sum().backward()
        torch.cuda.synchronize()
        assert torch.equal(outs[0].detach(), ref[0][0]) and int(outs[0].grad_fn.num_rendered) == R
    finally:
        dgr.set_async_forward(prev)


def test_non_blocking_forward_without_a_backward_is_resolved_by_the_next_forward_of_its_size():
    """a run-ahead forward that no backward follows (a visualisation render outside no_grad) used to keep its count — and
    a possible overflow, with all-NaN outputs — to itself for ever, and never corrected the binning hint.  The next run-ahead forward
    of the same (device, P, W, H) now resolves it first: the overflow is raised there, once, the hint is corrected, and the call after
    that fits.  LazyRendered also answers bool() and subtraction like the int it stands for."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    W, H, P, K = 120, 90, 1500, 11
    cam, sc, up = scenes.build(W, H, P, K, seed=8, kind="slam")
    key = (dev.index, P, W, H)
    ref = _fwd_bwd(cam, sc, up, dev)
    prev = dgr.set_async_forward(True)
    try:
        R = _C._binning_hint[key]
        leaf, outs = _render_sem(cam, sc, dev)                     # runs ahead, fits, nobody calls backward
        lz = outs[0].grad_fn.num_rendered
        assert isinstance(lz, _C.LazyRendered) and lz._value is None
        leaf, outs_b = _render_sem(cam, sc, dev)                   # the next forward of this size resolved it
        assert lz._value == R and bool(lz) and lz - 1 == R - 1 and R - lz == 0
        outs_b[0].sum().backward()
        _C._binning_hint[key] = 8                                  # now one that overflows and is never followed by a backward
        leaf, outs_c = _render_sem(cam, sc, dev)
        torch.cuda.synchronize()
        assert bool(torch.isnan(outs_c[0]).all())
        with pytest.raises(RuntimeError, match="does not fit the binning buffer"):
            _render_sem(cam, sc, dev)                              # raised HERE, at the latest
        assert _C._binning_hint[key] == R
        leaf, outs_d = _render_sem(cam, sc, dev)                   # not raised twice; this one fits
        outs_d[0].sum().backward()
        torch.cuda.synchronize()
        assert torch.equal(outs_d[0].detach(), ref[0][0])
        with pytest.raises(RuntimeError, match="does not fit the binning buffer"):
            int(outs_c[0].grad_fn.num_rendered)                    # whoever asks the failed one again gets the same answer
    finally:
        dgr.set_async_forward(prev)


def test_non_blocking_count_overwritten_after_a_ring_of_unresolved_forwards_fails_at_once():
    """The non-blocking forward's counts live in a ring of 256 host-mapped slots per device.  A forward whose count nobody resolved
    while 256 later ones ran ahead has lost it: resolving it then must raise immediately (not wait 10 s for a sequence number that
    will never come back), later forwards are unaffected."""
    import time
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    W, H, P, K = 64, 48, 400, 5
    cam, sc, up = scenes.build(W, H, P, K, seed=3, kind="slam")
    _fwd_bwd(cam, sc, up, dev)                        # blocking; leaves the binning hint of this size
    prev = dgr.set_async_forward(True)
    try:
        leaf0, outs0 = _render_sem(cam, sc, dev)      # never resolved ...
        first = outs0[0].grad_fn.num_rendered
        assert isinstance(first, _C.LazyRendered)
        # ... while a whole ring of later forwards OF ANOTHER SIZE runs ahead (each resolved by its backward).  (A later run-ahead forward
        # of the SAME size resolves the pending count first, since round 4: see the next test.)
        cam2, sc2, up2 = scenes.build(W, H, P + 1, K, seed=4, kind="slam")
        _fwd_bwd(cam2, sc2, up2, dev)
        for _ in range(260):
            leaf, outs = _render_sem(cam2, sc2, dev)
            assert isinstance(outs[0].grad_fn.num_rendered, _C.LazyRendered)
            outs[0].sum().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match="overwritten"):
            outs0[0].sum().backward()
        assert time.perf_counter() - t0 < 2.0
        leaf, outs = _render_sem(cam, sc, dev)        # the ring itself is fine
        outs[0].sum().backward()
        torch.cuda.synchronize()
        assert int(outs[0].grad_fn.num_rendered) > 0 and torch.isfinite(leaf["means3D"].grad).all()
    finally:
        dgr.set_async_forward(prev)


@pytest.mark.parametrize("W,H", [(1, 1), (3, 5), (16, 16), (17, 1), (1, 33), (15, 31)])
def test_tiny_and_sliver_images(W, H):
    """images smaller than a tile, exactly one tile, one pixel wide / high: ragged last tiles on both axes, sub-blocks entirely outside"""
    cam, sc, up = scenes.build(W, H, 120, 6, seed=W * 100 + H, kind="aniso", scale_mult=4.0, bg=(0.2, 0.0, 0.1))
    _compare(cam, sc, up, True, "sr", None)
    cam, sc, up = scenes.build(W, H, 40, 0, seed=W * 100 + H + 1, kind="slam", scale_mult=6.0)
    _compare(cam, sc, up, False, "sr", None)


def test_non_finite_gaussians_are_culled_and_harm_nothing():
    """NaN positions / scales / rotations make a Gaussian's projected covariance NaN; like the reference's, the preprocess then ends with
    radius 0 (ceil(NaN) converts to 0, the tile rectangle is empty) and the Gaussian takes no part: the render equals the render of the
    map without those Gaussians bit for bit, their gradients are exactly zero, nobody else's are touched — and nothing hangs or faults."""
    dev = torch.device("cuda:0")
    W, H, P, K = 150, 90, 2500, 11
    cam, sc, up = scenes.build(W, H, P, K, seed=21, kind="aniso", scale_mult=2.0)
    g = torch.Generator().manual_seed(4)
    bad = torch.rand(P, generator=g) < 0.04
    which = torch.randint(0, 3, (P,), generator=g)
    poisoned = {n: v.clone() for n, v in sc.items()}
    nan = float("nan")
    poisoned["means3D"][bad & (which == 0)] = nan
    poisoned["scales"][bad & (which == 1), 1] = nan
    poisoned["rotations"][bad & (which == 2), 2] = nan
    clean = {n: (v[~bad].clone() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == P else v) for n, v in sc.items()}
    lp, op = _render_sem(cam, poisoned, dev)
    lc, oc = _render_sem(cam, clean, dev)
    upd = {n: v.to(dev) for n, v in up.items()}
    for outs in (op, oc):
        color, radii, sem, depth, median, opac = outs
        ((color * upd["color"]).sum() + (sem * upd["semantic"]).sum() + (depth * upd["depth"]).sum() + (median * upd["median"]).sum()
         + (opac * upd["opacity"]).sum()).backward()
    torch.cuda.synchronize()
    badd = bad.to(dev)
    assert int((op[1][badd] != 0).sum()) == 0 and torch.equal(op[1][~badd], oc[1])            # radii
    for a, b in zip((op[0], op[2], op[3], op[4], op[5]), (oc[0], oc[2], oc[3], oc[4], oc[5])):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    for n in lp:
        gp, gc = lp[n].grad, lc[n].grad
        assert torch.isfinite(gp).all(), n
        assert not bool(gp[badd].any()), n
        _same_up_to_atomics_order({n: gc}, {n: gp[~badd]})


def test_no_device_or_host_memory_growth_over_many_steps():
    """Two rounds of 3 x 100 forward+backward steps (blocking forward, non-blocking forward, stage timers on): the first round brings every
    path's buffers and caches into being, over the second one device memory allocated through torch and the process's resident set stay
    put — no per-call leak in the glue, the library's event pool, the host-mapped counters or the ticket ring."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C
    dev = torch.device("cuda:0")
    W, H, P, K = 160, 96, 3000, 16
    cam, sc, up = scenes.build(W, H, P, K, seed=8, kind="slam")
    upd = {n: v.to(dev) for n, v in up.items()}

    def rss_mb():
        with open("/proc/self/statm") as f:
            return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2 ** 20

    def step():
        leaf, outs = _render_sem(cam, sc, dev)
        color, radii, sem, depth, median, opac = outs
        ((color * upd["color"]).sum() + (sem * upd["semantic"]).sum() + (depth * upd["depth"]).sum()).backward()

    def one_round():
        for phase in range(3):
            prev = dgr.set_async_forward(phase == 1)
            if phase == 2:
                _C._lib.hsr_profile_enable(1)
            try:
                for _ in range(100):
                    step()
                torch.cuda.synchronize()
            finally:
                dgr.set_async_forward(prev)
                if phase == 2:
                    _C._lib.hsr_profile_read(None, 1)
                    _C._lib.hsr_profile_enable(0)
    one_round()
    torch.cuda.synchronize()
    mem0, rss0 = torch.cuda.memory_allocated(dev), rss_mb()
    one_round()
    one_round()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated(dev) <= mem0 + (1 << 20), (torch.cuda.memory_allocated(dev), mem0)
    assert rss_mb() - rss0 < 64.0, (rss_mb(), rss0)


def test_two_host_threads_render_concurrently():
    """two Python threads, each with its own stream, scene and image size, 40 forward+backward steps each at the same time: every thread
    gets exactly the images (and, within atomics noise, the gradients) it gets alone — the library's per-(thread, device) read-back
    slots, hints and error strings do not bleed between callers"""
    import threading
    dev = torch.device("cuda:0")
    jobs = [(203, 131, 3000, 26, 5, "slam"), (96, 150, 1800, 5, 6, "aniso")]
    scenes_ = [scenes.build(W, H, P, K, seed=s, kind=kind) for (W, H, P, K, s, kind) in jobs]
    alone = [_fwd_bwd(cam, sc, up, dev) for (cam, sc, up) in scenes_]
    results, errors = [None, None], []

    def work(i):
        try:
            cam, sc, up = scenes_[i]
            st = torch.cuda.Stream(dev)
            with torch.cuda.stream(st):
                for _ in range(40):
                    r = _fwd_bwd(cam, sc, up, dev)
                st.synchronize()
            results[i] = r
        except Exception as e:   # noqa: BLE001
            errors.append((i, repr(e)))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for ref, got in zip(alone, results):
        assert got is not None
        for a, b in zip(ref[0], got[0]):
            assert torch.equal(a, b)
        assert torch.equal(ref[1], got[1])
        _same_up_to_atomics_order(ref[2], got[2])


@pytest.mark.parametrize("mod", [0.6, 1.7])
def test_scale_modifier_other_than_one(mod):
    """raster_settings.scale_modifier multiplies every scale before the covariance (forward.cu:118-124) and its gradient (backward.cu:
    295-300); Hier-SLAM always passes 1.0 (utils/recon_helpers.py:20), the API allows anything"""
    cam, sc, up = scenes.build(150, 90, 2000, 11, seed=31, kind="aniso", scale_mult=2.0)
    cam = dict(cam, scale_modifier=float(mod))
    _compare(cam, sc, up, True, "sr", None)
    cam2, sc2, up2 = scenes.build(96, 80, 1200, 0, seed=32, kind="aniso", scale_mult=2.0)
    _compare(dict(cam2, scale_modifier=float(mod)), sc2, up2, False, "sr", None)


def test_off_centre_principal_point_and_unequal_focal_lengths():
    """ScanNet-like intrinsics: fx != fy and (cx, cy) away from the image centre — tanfovx / tanfovy then differ and the projection
    matrix is asymmetric (utils/recon_helpers.py:4-28 builds both from the same K); the scenes of the other cases all use centred
    Replica intrinsics"""
    import numpy as np
    from hsr_utils.camera import setup_camera_tensors
    from hsr_utils.synthetic import make_scene, make_upstream_grads
    W, H, P, K = 160, 120, 2500, 16
    k = np.array([[155.0, 0.0, 0.41 * W], [0.0, 171.0, 0.57 * H], [0.0, 0.0, 1.0]])
    w2c = scenes.tilted_w2c(0.15, (0.05, 0.1, 0.1))
    cam = setup_camera_tensors(W, H, k, w2c)
    assert abs(cam["tanfovx"] * 155.0 / W - cam["tanfovy"] * 171.0 / H) < 1e-6 and cam["tanfovx"] / cam["tanfovy"] != W / H
    sc = make_scene(P, W, H, K, k, seed=12, kind="aniso", scale_mult=2.0, w2c=w2c)
    up = {n: v * float(W * H) for n, v in make_upstream_grads(W, H, K, seed=2).items()}
    _compare(cam, sc, up, True, "sr", None)


@pytest.mark.parametrize("glue", ["compiled", "ctypes"])
def test_non_contiguous_inputs_and_partial_requires_grad(glue):
    """inputs that are strided views (a column slice of a wider table, a transposed buffer) render like their contiguous copies, bit
    for bit; gradients come back for exactly the inputs that asked for one"""
    import subprocess
    import sys
    code = r'''
import sys; sys.path[:0]=['hier-slam_amd','tests']
import torch, scenes
from diff_gaussian_rasterization import GaussianRasterizer_semantic
from harness import _cam_to
dev = torch.device("cuda:0")
cam, sc, up = scenes.build(150, 90, 2000, 11, seed=41, kind="aniso", scale_mult=2.0)
P = sc["means3D"].shape[0]
def render(inputs):
    return GaussianRasterizer_semantic(_cam_to(cam, dev))(means2D=torch.zeros(P, 3, device=dev, requires_grad=True), **inputs)
names = ("means3D", "opacities", "colors_precomp", "scales", "rotations", "semantics_precomp")
plain = {n: sc[n].to(dev).clone().requires_grad_(n in ("means3D", "colors_precomp")) for n in names}
strided = {}
for n in names:
    t = sc[n].to(dev)
    wide = torch.zeros(P, t.shape[1] + 3, device=dev); wide[:, 1:1 + t.shape[1]] = t
    v = wide[:, 1:1 + t.shape[1]] if n != "rotations" else t.t().contiguous().t()      # column slice / transposed storage
    assert not v.is_contiguous()
    strided[n] = v.detach().requires_grad_(n in ("means3D", "colors_precomp"))
oa, ob = render(plain), render(strided)
for a, b in zip(oa, ob):
    assert torch.equal(a, b)
upd = {n: v.to(dev) for n, v in up.items()}
for outs in (oa, ob):
    color, radii, sem, depth, median, opac = outs
    ((color * upd["color"]).sum() + (sem * upd["semantic"]).sum() + (depth * upd["depth"]).sum()).backward()
for n in names:
    if n in ("means3D", "colors_precomp"):
        ga, gb = plain[n].grad, strided[n].grad
        assert ga is not None and gb is not None and gb.shape == ga.shape
        assert float((ga - gb).abs().max()) <= (1e-3 if n == "means3D" else 2e-5) * max(1.0, float(ga.abs().max())), n
    else:
        assert plain[n].grad is None and strided[n].grad is None, n
print("ok")
'''
    env = dict(os.environ)
    if glue == "ctypes":
        env["HSR_GLUE"] = "ctypes"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_backward_twice_through_a_retained_graph():
    """loss.backward(retain_graph=True) twice: the saved state buffers serve both passes (the reference's do too) and the leaves'
    gradients accumulate to exactly twice one pass's (up to the atomics' order); also with the non-blocking forward, whose count is
    resolved by the first pass and reused by the second"""
    import diff_gaussian_rasterization as dgr
    dev = torch.device("cuda:0")
    cam, sc, up = scenes.build(150, 90, 2000, 11, seed=43, kind="slam")
    upd = {n: v.to(dev) for n, v in up.items()}
    once = _fwd_bwd(cam, sc, up, dev)[2]
    for asyn in (False, True):
        prev = dgr.set_async_forward(asyn)
        try:
            leaf, outs = _render_sem(cam, sc, dev)
            color, radii, sem, depth, median, opac = outs
            loss = (color * upd["color"]).sum() + (sem * upd["semantic"]).sum() + (depth * upd["depth"]).sum() \
                + (median * upd["median"]).sum() + (opac * upd["opacity"]).sum()
            loss.backward(retain_graph=True)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            dgr.set_async_forward(prev)
        _same_up_to_atomics_order({n: 2.0 * v for n, v in once.items()}, {n: leaf[n].grad for n in leaf})


@pytest.mark.parametrize("used", [("depth", "color"), ("semantic",), ("opacity",)])
def test_outputs_the_loss_does_not_use_cost_no_zero_filled_maps(used):
    """autograd would hand backward() a freshly zero-filled map for every output the loss does not touch (a K x H x W fill per tracking
    iteration); the node asks for None instead and substitutes cached, never-written zero maps.  Gradients = the oracle's with zero upstream
    gradients for the unused outputs; the cached maps are still all zero afterwards."""
    import diff_gaussian_rasterization as D
    from diff_gaussian_rasterization import GaussianRasterizer_semantic
    from harness import _cam_to
    W, H, P, K = 128, 80, 2000, 16
    cam, sc, up = scenes.build(W, H, P, K, seed=13, kind="slam", scale_mult=3.0)
    dev = torch.device("cuda:0")
    leaf = {n: sc[n].to(dev).clone().requires_grad_(True) for n in ("means3D", "opacities", "colors_precomp", "scales", "rotations", "semantics_precomp")}
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, radii, sem, depth, median, opacity = GaussianRasterizer_semantic(_cam_to(cam, dev))(means2D=means2D, **leaf)
    outs = dict(color=color, semantic=sem, depth=depth, median=median, opacity=opacity)
    loss = sum((outs[n] * up[n].to(dev)).sum() for n in used)
    loss.backward()
    torch.cuda.synchronize()
    up0 = {n: (v if n in used else torch.zeros_like(v)) for n, v in up.items()}
    _, go, so = run_oracle(cam, sc, up0, semantic=True, variant="sr")
    import harness
    for n in leaf:
        g = leaf[n].grad.cpu().numpy()
        assert_close("grad %s (loss on %s only)" % (n, "+".join(used)), g, go[n], allowance=harness.tie_allowance("grad " + n, so, g.shape, "gauss"))
    so.free()
    cached = [z for (d, shape), z in D._zero_maps.items() if d == dev]
    assert cached, "no cached zero map was used"
    for z in cached:
        assert not z.any(), "a cached zero map was written to"
