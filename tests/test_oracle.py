"""CPU suite: pins the C oracle.

The reference holds no tests or golden vectors for this path and is CUDA-only, so the
oracle is "parity unpinned" against reference OUTPUTS.  What pins it instead:
  1. an independent float64 torch.autograd restatement of the differentiable maths (tests/dense_ref.py),
  2. algebraic identities of alpha compositing,
  3. committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from this oracle)
     that freeze its behaviour, so later edits cannot drift silently.
"""
import os

import numpy as np
import pytest
import torch

import dense_ref as DR
import oracle_lib as O
import scenes
from harness import run_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64).reshape(np.asarray(a).shape)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


@pytest.mark.parametrize("kind,bg", [("aniso", (0.3, 0.6, 0.1)), ("slam", (0, 0, 0))])
def test_oracle_exact_semantic_alpha_mode_matches_float64_autograd(kind, bg):
    """sem_alpha_mode 1 ("exact": the semantic loss also reaches alpha, what backward.cu:778-779 / :834-845 intended; the product's opt-in
    `set_semantic_alpha("exact")`): pinned by the same float64 autograd restatement with the term kept (w not detached in the semantic sums).
    It must differ from the default mode wherever a semantic gradient flows, and leave dL_dsemantics itself unchanged."""
    W, H, K, P = 40, 36, 5, 60
    cam, sc, up = scenes.build(W, H, P, K, seed=3, kind=kind, scale_mult=4.0, bg=bg)
    out, gr0, st = run_oracle(cam, sc, up, semantic=True, threads=2)
    g = {n: v.numpy() for n, v in up.items()}
    kw = dict(colors_precomp=sc["colors_precomp"], semantics_precomp=sc["semantics_precomp"], scales=sc["scales"], rotations=sc["rotations"])
    gr = O.backward(st, cam, sc["means3D"], g, threads=2, median_rule="forward", sem_alpha_exact=True, **kw)
    dout, dgr = DR.dense_loss_and_grads(cam, sc, g, st.field("vals"), st.field("ranges"), semantic=True, sem_alpha_exact=True)
    for a, b in dict(means3D="means3D", scales="scales", rotations="rotations", opacities="opacities_ref", colors_precomp="colors",
                     semantics_precomp="semantics").items():
        assert _rel(gr[a], dgr[b].numpy()) < 2e-5, (a, _rel(gr[a], dgr[b].numpy()))
    assert _rel(gr["means2D"][:, :2], dgr["ndc_delta"].numpy()) < 2e-5
    assert _rel(gr["semantics_precomp"], gr0["semantics_precomp"]) < 1e-12          # sum w * dL_dsem: the same in both modes
    assert _rel(gr["opacities"], gr0["opacities"]) > 1e-3 and _rel(gr["means3D"], gr0["means3D"]) > 1e-3   # the term is not small
    st.free()


@pytest.mark.parametrize("kind,semantic,bg", [("aniso", True, (0.3, 0.6, 0.1)), ("slam", True, (0, 0, 0)),
                                              ("aniso", False, (0.0, 0.0, 0.0))])
def test_oracle_matches_float64_autograd(kind, semantic, bg):
    W, H, K, P = 40, 36, 5, 60
    cam, sc, up = scenes.build(W, H, P, K if semantic else 0, seed=3, kind=kind, scale_mult=4.0, bg=bg)
    out, gr, st = run_oracle(cam, sc, up, semantic=semantic, threads=2)
    g = {n: v.numpy() for n, v in up.items()}
    dout, dgr = DR.dense_loss_and_grads(cam, sc, g, st.field("vals"), st.field("ranges"), semantic=semantic)
    assert int((st.field("n_contrib").reshape(H, W) != dout["n_contrib"].numpy()).sum()) == 0
    for a, b in (("color", "color"), ("depth", "depth"), ("median_depth", "median"), ("opacity", "opacity")):
        assert np.abs(out[a] - dout[b].detach().numpy()).max() < 5e-6, a
    if semantic:
        assert np.abs(out["semantic"] - dout["semantic"].detach().numpy()).max() < 5e-6
    else:
        assert np.abs(out["mask"] - dout["mask"].detach().numpy()).max() < 5e-6
    pairs = dict(means3D="means3D", scales="scales", rotations="rotations", opacities="opacities_ref", colors_precomp="colors")
    if semantic:
        pairs["semantics_precomp"] = "semantics"
    for a, b in pairs.items():
        assert _rel(gr[a], dgr[b].numpy()) < 2e-5, (a, _rel(gr[a], dgr[b].numpy()))
    assert _rel(gr["means2D"][:, :2], dgr["ndc_delta"].numpy()) < 2e-5
    st.free()


def test_oracle_cov3d_precomp_matches_autograd():
    W, H, K, P = 40, 36, 4, 50
    cam, sc, up = scenes.build(W, H, P, K, seed=8, kind="aniso", scale_mult=4.0)
    sc["cov3D_precomp"] = scenes.cov3d_from_scene(sc)
    out, gr, st = run_oracle(cam, sc, up, semantic=True, variant="cov", extra=sc, threads=2)
    g = {n: v.numpy() for n, v in up.items()}
    dout, dgr = DR.dense_loss_and_grads(cam, sc, g, st.field("vals"), st.field("ranges"), use_cov3d=True)
    # off-diagonal entries appear twice in the symmetric matrix: the reference reports the full derivative
    # w.r.t. the 6 stored numbers (backward.cu:221-227), which is what autograd on the 6-vector gives
    assert _rel(gr["cov3D_precomp"], dgr["cov3D"].numpy()) < 2e-5
    assert _rel(gr["means3D"], dgr["means3D"].numpy()) < 2e-5
    st.free()


def test_compositing_identities():
    """sum_g dL_ddepth[g] == sum_pix opacity and sum dL_dsemantics == K * sum opacity for all-ones upstream"""
    W, H, K, P = 96, 64, 26, 400
    cam, sc, _ = scenes.build(W, H, P, K, seed=0, kind="aniso", scale_mult=3.0)
    kw = dict(colors_precomp=sc["colors_precomp"], semantics_precomp=sc["semantics_precomp"], scales=sc["scales"],
              rotations=sc["rotations"])
    out, st = O.forward(cam, sc["means3D"], sc["opacities"], threads=2, **kw)
    ones = lambda c: np.ones((c, H, W), np.float32)
    zeros = lambda c: np.zeros((c, H, W), np.float32)
    g = dict(color=zeros(3), semantic=ones(K), depth=ones(1), median=zeros(1), opacity=zeros(1))
    gr = O.backward(st, cam, sc["means3D"], g, threads=2, **kw)
    tot = float(out["opacity"].sum())
    assert abs(gr["depths"].sum() - tot) < 1e-3 * tot
    assert abs(gr["semantics_precomp"].sum() - K * tot) < 1e-3 * K * tot
    # opacity = 1 - final_T and final_T in (0, 1]
    fT = st.field("final_T")
    assert np.allclose(out["opacity"].reshape(-1), 1 - fT, atol=1e-7) and fT.min() > 0 and fT.max() <= 1
    # sortedness and range consistency of the binning
    keys = st.field("keys")
    assert np.all(keys[1:] >= keys[:-1])
    rg = st.field("ranges")
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    for t in np.unique(tiles):
        idx = np.nonzero(tiles == t)[0]
        assert rg[t, 0] == idx[0] and rg[t, 1] == idx[-1] + 1
    assert int((rg[:, 1] - rg[:, 0]).sum()) == st.R
    st.free()


def test_stable_sort_ties_keep_gaussian_order():
    """identical Gaussians -> identical (tile, depth) keys: the sorted values must stay in ascending index order"""
    W, H, K, P = 48, 32, 3, 40
    cam, sc, _ = scenes.build(W, H, P, K, seed=2, kind="slam", scale_mult=3.0, tilt=False)
    for n in ("means3D", "scales", "rotations"):
        sc[n][1::2] = sc[n][0::2]  # pairs of coincident Gaussians
    kw = dict(colors_precomp=sc["colors_precomp"], semantics_precomp=sc["semantics_precomp"], scales=sc["scales"],
              rotations=sc["rotations"])
    out, st = O.forward(cam, sc["means3D"], sc["opacities"], **kw)
    keys, vals = st.field("keys"), st.field("vals").astype(np.int64)
    same = keys[1:] == keys[:-1]
    assert same.any()
    assert np.all(vals[1:][same] > vals[:-1][same])
    st.free()


def test_get_higher_msb():
    # reference getHigherMsb (rasterizer_impl.cu:35-50): 3225 tiles -> 12 bits, 8160 -> 13
    assert O.get_higher_msb(3225) == 12 and O.get_higher_msb(8160) == 13
    assert O.get_higher_msb(1) == 1 and O.get_higher_msb(4096) == 13 and O.get_higher_msb(4095) == 12


def test_no_gaussians_and_all_culled():
    cam, sc, up = scenes.build(32, 32, 0, 4)
    out, st = O.forward(cam, sc["means3D"].reshape(0, 3), sc["opacities"].reshape(0, 1),
                        colors_precomp=np.zeros((0, 3), np.float32), semantics_precomp=np.zeros((0, 4), np.float32),
                        scales=np.zeros((0, 3), np.float32), rotations=np.zeros((0, 4), np.float32))
    assert out["num_rendered"] == 0 and float(np.abs(out["color"]).max()) == 0
    assert float(out["median_depth"].min()) == 15.0  # the oracle renders empty tiles: default median (forward.cu:450)
    st.free()


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(GOLD) if f.endswith(".npz") and not f.startswith(("loss_", "densify_"))) if os.path.isdir(GOLD) else [])
def test_golden_fixture(name):
    """the oracle reproduces the committed fixtures bit-for-bit on integers and to 1e-6 on floats"""
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    cam = dict(image_height=int(z["H"]), image_width=int(z["W"]), tanfovx=float(z["tanfovx"]), tanfovy=float(z["tanfovy"]),
               bg=torch.tensor(z["bg"]), scale_modifier=float(z["scale_modifier"]), viewmatrix=torch.tensor(z["viewmatrix"]),
               projmatrix=torch.tensor(z["projmatrix"]), sh_degree=int(z["sh_degree"]), campos=torch.tensor(z["campos"]),
               prefiltered=False, debug=False)
    semantic = bool(z["semantic"])
    kw = dict(colors_precomp=z["colors_precomp"], scales=z["scales"], rotations=z["rotations"])
    if semantic:
        kw["semantics_precomp"] = z["semantics_precomp"]
    out, st = O.forward(cam, z["means3D"], z["opacities"], threads=2, **kw)
    assert out["num_rendered"] == int(z["exp_num_rendered"])
    for n in ("radii",):
        assert np.array_equal(out[n], z["exp_" + n]), n
    for n in ("keys", "vals", "ranges", "tiles_touched", "n_contrib"):
        assert np.array_equal(st.field(n), z["exp_" + n]), n
    for n in ("color", "depth", "median_depth", "opacity") + (("semantic",) if semantic else ("mask",)):
        assert np.abs(out[n] - z["exp_" + n]).max() <= 1e-6, n
    g = dict(color=z["up_color"], semantic=z["up_semantic"] if semantic else None, depth=z["up_depth"], median=z["up_median"],
             opacity=z["up_opacity"])
    gr = O.backward(st, cam, z["means3D"], g, threads=2, **kw)
    for n in ("means3D", "means2D", "opacities", "colors_precomp", "scales", "rotations") + (("semantics_precomp",) if semantic else ()):
        assert _rel(gr[n], z["exp_grad_" + n]) < 1e-6, n
    st.free()


# ---- spherical-harmonics colour path: pinned independently of the oracle's polynomials ----------------

def test_real_sh_basis_matches_scipy():
    """tests/dense_ref.real_sh_basis (general definition) against scipy.special's complex harmonics, bands 0-3"""
    from scipy import special
    rng = np.random.default_rng(0)
    d = rng.normal(size=(64, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    Y = DR.real_sh_basis(torch.as_tensor(d), 3).numpy()
    theta = np.arccos(np.clip(d[:, 2], -1, 1))          # polar
    phi = np.arctan2(d[:, 1], d[:, 0])                   # azimuth
    harm = getattr(special, "sph_harm_y", None)
    col = 0
    for l in range(4):
        for m in range(-l, l + 1):
            a = abs(m)
            c = harm(l, a, theta, phi) if harm is not None else special.sph_harm(a, l, phi, theta)
            exp = c.real if m == 0 else np.sqrt(2.0) * (c.real if m > 0 else c.imag)
            assert np.abs(Y[:, col] - exp).max() < 1e-12, (l, m)
            col += 1
    # and the reference's published constants (auxiliary.h:22-39) are these harmonics' leading coefficients
    assert abs(Y[0, 0] - 0.28209479177387814) < 1e-15


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_sh_matches_float64_autograd(deg):
    """computeColorFromSH forward (forward.cu:20-71) and backward (backward.cu:20-139) of the oracle against the
    float64 autograd restatement built on the general harmonics; some colours are driven negative so that the clamp
    flags and their zero gradient are exercised."""
    W, H, K, P = 40, 36, 4, 60
    cam, sc, up = scenes.build(W, H, P, K, seed=4, kind="aniso", scale_mult=4.0)
    cam["sh_degree"] = deg
    shs = scenes.random_sh(P, 16, seed=9)
    shs[::5, 0, :] -= 3.2                       # -> negative before the clamp on every 5th Gaussian
    shs[1::7, 0, 1] -= 3.2                      # and single channels
    extra = {"shs": shs}
    out, gr, st = run_oracle(cam, sc, up, semantic=True, extra=extra, threads=2)
    g = {n: v.numpy() for n, v in up.items()}
    dout, dgr = DR.dense_loss_and_grads_sh(cam, sc, shs.numpy(), deg, g, st.field("vals"), st.field("ranges"))
    vis = out["radii"] > 0
    rgb = st.field("rgb")
    assert np.abs(rgb[vis] - dgr["colors_value"].numpy()[vis]).max() < 2e-6
    clamped = st.field("clamped").astype(bool)
    assert clamped[vis].any() and not clamped[vis].all()
    assert np.array_equal(clamped[vis], (dgr["colors_value"].numpy()[vis] == 0))
    assert np.abs(out["color"] - dout["color"].detach().numpy()).max() < 5e-6
    nb = (deg + 1) ** 2
    assert _rel(gr["shs"][:, :nb], dgr["shs"].numpy()[:, :nb]) < 2e-5
    assert float(np.abs(gr["shs"][:, nb:]).max()) == 0 if nb < 16 else True
    assert float(np.abs(gr["shs"][clamped[:, 0], :, 0]).max()) == 0   # clamped channel: no gradient (backward.cu:33-38)
    for a, b in dict(means3D="means3D", scales="scales", rotations="rotations", opacities="opacities_ref",
                     semantics_precomp="semantics").items():
        assert _rel(gr[a], dgr[b].numpy()) < 2e-5, (a, _rel(gr[a], dgr[b].numpy()))
    st.free()


def _oracle_loss(cam, sc, up, extra=None):
    """L = sum(outputs * upstream) in float64 from one oracle forward"""
    kw = dict(scales=sc["scales"], rotations=sc["rotations"], semantics_precomp=sc["semantics_precomp"])
    if extra is not None:
        kw["shs"] = extra["shs"]
    else:
        kw["colors_precomp"] = sc["colors_precomp"]
    out, st = O.forward(cam, sc["means3D"], sc["opacities"], threads=2, **kw)
    st.free()
    L = 0.0
    for a, b in (("color", "color"), ("semantic", "semantic"), ("depth", "depth"), ("median_depth", "median"), ("opacity", "opacity")):
        L += float((out[a].astype(np.float64) * up[b].numpy().astype(np.float64)).sum())
    return L


@pytest.mark.parametrize("use_sh", [False, True])
def test_oracle_gradients_match_finite_differences(use_sh):
    """Central differences of the oracle's OWN forward against its analytic backward — no second derivation involved.

    What such a check can and cannot show.  The reference's backward is not the derivative of its forward in five places;
    four are switched off here: Gaussians outside 0.9 x the 1.3 tan(fov) frustum clamp are left out (the clamped
    coordinate gets gradient 0 by decree, backward.cu:175-176, while the forward does move with it), final opacity's upstream is 0 (its colour-style term is added to dL_dopacity,
    backward.cu:859-864), bg = 0 (the forward never composites it, forward.cu:530-531), the semantic upstream is 0 except
    when dL_dsemantics itself is checked (the semantic loss never reaches alpha, backward.cu:834), and median depth
    (piecewise constant) gets upstream 0.  The fifth cannot be switched off: the analytic gradient treats the support of a
    splat (alpha >= 1/255) as fixed, while a finite difference sees the rim move — for a splat of opacity o that is a
    systematic (2 ln(255 o)/255)/(2 o) of the main term, 2-3 % at o > 0.8 — so the geometric parameters are compared at
    3 % of the tensor's largest gradient on a scene of opaque splats and smooth upstream fields (white-noise upstreams
    turn every rim pixel into a +-|up|/255/h jump); colours, SH coefficients and semantics enter linearly and are compared
    at 2e-3."""
    W, H, K, P = 40, 36, 4, 50
    cam, sc, up = scenes.build(W, H, P, K, seed=6, kind="aniso", scale_mult=5.0)
    sc["opacities"] = (0.8 + 0.19 * torch.rand(P, 1, generator=torch.Generator().manual_seed(2))).float()
    extra = None
    if use_sh:
        cam["sh_degree"] = 3
        extra = {"shs": scenes.random_sh(P, 16, seed=3)}
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    smooth = lambda a, b, c: (a + b * xx + c * yy + 0.5 * xx * yy).float()[None].contiguous()
    up = dict(color=torch.cat([smooth(0.6, 0.5, -0.3), smooth(-0.4, 0.2, 0.6), smooth(0.3, -0.6, 0.2)]).contiguous(),
              semantic=torch.zeros(K, H, W), depth=smooth(0.2, -0.3, 0.4), median=torch.zeros(1, H, W),
              opacity=torch.zeros(1, H, W))
    up_sem = torch.cat([smooth(0.1 * k, 0.3, -0.2 * k) for k in range(K)]).contiguous()
    _, gr, st = run_oracle(cam, sc, up, semantic=True, extra=extra, threads=2)
    st.free()
    rng = np.random.default_rng(1)
    t = torch.cat([sc["means3D"], torch.ones(P, 1)], 1) @ cam["viewmatrix"].reshape(4, 4)
    inside = ((t[:, 0] / t[:, 2]).abs() < 0.9 * 1.3 * cam["tanfovx"]) & ((t[:, 1] / t[:, 2]).abs() < 0.9 * 1.3 * cam["tanfovy"])
    inside = inside.numpy()
    assert 10 < inside.sum() < P
    params = [("means3D", sc["means3D"], 0.03), ("scales", sc["scales"], 0.03), ("rotations", sc["rotations"], 0.03),
              ("opacities", sc["opacities"], 0.03)]
    params.append(("shs", extra["shs"], 2e-3) if use_sh else ("colors_precomp", sc["colors_precomp"], 2e-3))
    checked, worst = 0, {}
    for name, tens, tol in params:
        ga = gr[name].reshape(tens.shape)
        scale = float(np.abs(ga[inside]).max())
        gsel = np.where(inside.reshape((-1,) + (1,) * (ga.ndim - 1)), np.abs(ga), 0.0)
        flat = np.argsort(-gsel.reshape(-1))[:40]                # entries with a large analytic gradient
        for idx in rng.choice(flat, size=6, replace=False):
            ix = tuple(int(i) for i in np.unravel_index(int(idx), tens.shape))
            x0 = float(tens[ix])
            h = (0.25 if tol < 1e-2 else 4e-3 * max(abs(x0), 0.05))
            tens[ix] = x0 + h; Lp = _oracle_loss(cam, sc, up, extra)
            tens[ix] = x0 - h; Lm = _oracle_loss(cam, sc, up, extra)
            tens[ix] = x0
            fd = (Lp - Lm) / (2 * h)
            worst[name] = max(worst.get(name, 0.0), abs(fd - float(ga[ix])) / scale)
            assert abs(fd - float(ga[ix])) <= tol * scale + 1e-6, (name, ix, fd, float(ga[ix]), scale)
            checked += 1
    # dL_dsemantics: linear in the semantics, exact up to rounding
    up["semantic"] = up_sem
    _, gr, st = run_oracle(cam, sc, up, semantic=True, extra=extra, threads=2)
    st.free()
    ga = gr["semantics_precomp"]
    for idx in rng.choice(np.argsort(-np.abs(ga).reshape(-1))[:40], size=6, replace=False):
        ix = tuple(int(i) for i in np.unravel_index(int(idx), ga.shape))
        x0 = float(sc["semantics_precomp"][ix])
        sc["semantics_precomp"][ix] = x0 + 0.25; Lp = _oracle_loss(cam, sc, up, extra)
        sc["semantics_precomp"][ix] = x0 - 0.25; Lm = _oracle_loss(cam, sc, up, extra)
        sc["semantics_precomp"][ix] = x0
        e = abs((Lp - Lm) / 0.5 - float(ga[ix])) / float(np.abs(ga).max())
        worst["semantics"] = max(worst.get("semantics", 0.0), e)
        assert e <= 2e-3, ("semantics", ix)
        checked += 1
    print("finite differences, worst |fd - analytic| / max|analytic|:", {k: "%.2e" % v for k, v in worst.items()})
    assert checked == 36


def test_scale_rotation_gradients_are_the_ill_conditioned_ones():
    """Why tests/harness.py holds dL_dscales / dL_drotations to a higher element-wise floor than the other gradients: on the
    oracle ALONE, a 1e-6 relative perturbation of the upstream gradients (the size of fp32 summation-order noise) moves
    those two several times further, relative to max(|g_i|, 0.1 max|g|), than any other tensor — the conic -> cov2D ->
    cov3D -> scale / quaternion chain (backward.cu:196-341) amplifies it.  Measured here, not assumed."""
    W, H, P, K = 136, 141, 2500, 8
    cam, sc, up = scenes.build(W, H, P, K, seed=455, kind="aniso", scale_mult=3.0, bg=(0.3, 0.0, 1.0))
    _, g0, st = run_oracle(cam, sc, up, semantic=True)
    st.free()
    rng = np.random.default_rng(0)
    worst = {}
    for _ in range(2):
        up2 = {n: v * (1 + 1e-6 * torch.tensor(rng.standard_normal(tuple(v.shape)), dtype=torch.float32)) for n, v in up.items()}
        _, g1, st = run_oracle(cam, sc, up2, semantic=True)
        st.free()
        for n in g0:
            mx = float(np.abs(g0[n]).max())
            if mx > 0:
                r = float((np.abs(g1[n] - g0[n]) / np.maximum(np.abs(g0[n]), 0.1 * mx)).max())
                worst[n] = max(worst.get(n, 0.0), r)
    print("amplification of a 1e-6 perturbation at floor 0.1:", {n: "%.0fx" % (v / 1e-6) for n, v in worst.items()})
    others = max(v for n, v in worst.items() if n not in ("scales", "rotations"))
    assert max(worst["scales"], worst["rotations"]) > 2.0 * others
    assert max(worst["scales"], worst["rotations"]) > 2e-5        # > 20x amplification


# ---- the truth build (libhsr_oracle_f64.so) and the tie bounds ------------------------------------------------------------

@pytest.mark.parametrize("cfg", [(96, 64, 800, 26, "aniso", 2.0), (170, 112, 2500, 74, "aniso", 3.0), (69, 39, 300, 1, "aniso", 1.0),
                                 (128, 80, 2000, 16, "slam", 3.0)])
def test_truth_build_walks_the_same_lists_and_the_oracle_is_one_rounding_from_it(cfg):
    """libhsr_oracle_f64.so runs the SAME fp32 preprocess / key / sort code (identical radii, keys, lists) and then the
    compositing, its backward and the per-Gaussian chain in double.  The fp32 oracle must sit within a few fp32 roundings of
    it everywhere — except where one of its threshold decisions fell within ulps of the threshold and the double evaluation
    took it the other way: there the difference must be covered by the bound the oracle computed for exactly that entry."""
    import harness
    W, H, P, K, kind, sm = cfg
    cam, sc, up = scenes.build(W, H, P, K, seed=11, kind=kind, scale_mult=sm)
    o32, g32, s32 = run_oracle(cam, sc, up, semantic=True, threads=4)
    o64, g64, s64 = run_oracle(cam, sc, up, semantic=True, threads=4, precision="f64", bounds=False)
    assert o64["color"].dtype == np.float64 and g64["means3D"].dtype == np.float64
    for n in ("radii",):
        assert np.array_equal(o32[n], o64[n])
    for n in ("keys", "vals", "ranges", "tiles_touched", "depths", "means2D", "conic_opacity"):
        assert np.array_equal(s32.field(n), s64.field(n)), n
    tie = s32.field("tie_pixels").astype(bool)
    # thresholded integers may differ on flagged pixels only
    assert not ((s32.field("n_contrib") != s64.field("n_contrib")) & ~tie).any()
    for n in ("color", "depth", "opacity", "semantic"):
        a = harness.tie_allowance(n, s32, o32[n].shape, "pixel")
        d = np.maximum(np.abs(o32[n].astype(np.float64) - o64[n]) - a, 0.0)
        assert d.max() <= 3e-6 * max(1.0, np.abs(o64[n]).max()), (n, d.max())
    for n in g32:
        t = np.asarray(g64[n], np.float64)
        a = harness.tie_allowance("grad " + n, s32, t.shape, "gauss")
        d = np.maximum(np.abs(np.asarray(g32[n], np.float64).reshape(t.shape) - t) - a, 0.0)
        mx = float(np.abs(t).max())
        assert d.max() <= 2e-5 * mx, (n, d.max() / mx)      # the cov chain amplifies one rounding ~50x (see the conditioning test)
    s32.free(); s64.free()


def test_tie_bound_covers_a_decision_forced_the_other_way():
    """One splat centred exactly on a pixel centre with opacity exactly 1/255: its alpha on that pixel IS the threshold (power = 0,
    exp = 1), so the oracle accepts it (alpha >= 1/255) and flags the decision.  Lowering the opacity by one ulp rejects it there:
    the two renders differ on that pixel by the splat's whole contribution and in the gradient rows of everything behind and in
    front of it.  The bound the first render computed must cover that difference, and must not be much larger than it."""
    import harness
    W, H, K, P = 48, 32, 5, 60
    cam, sc, up = scenes.build(W, H, P, K, seed=21, kind="aniso", scale_mult=6.0, tilt=False)
    # camera-centred splat j projected onto pixel (px, py): replica intrinsics scaled to W x H, w2c = I
    from hsr_utils.camera import replica_intrinsics
    k = replica_intrinsics(W, H)
    px, py, z = 20, 13, 1.5
    j = 7
    # pixel = f * X / Z + c - 0.5 (ndc2Pix of the OpenGL-style projection, auxiliary.h:43)
    sc["means3D"][j] = torch.tensor([(px + 0.5 - k[0, 2]) * z / k[0, 0], (py + 0.5 - k[1, 2]) * z / k[1, 1], z], dtype=torch.float32)
    sc["opacities"][j] = float(np.float32(1.0) / np.float32(255.0))
    outs = []
    for nudge in (False, True):
        s2 = {n: v.clone() for n, v in sc.items()}
        if nudge:
            s2["opacities"][j] = float(np.nextafter(np.float32(float(s2["opacities"][j])), np.float32(0)))
        outs.append(run_oracle(cam, s2, up, semantic=True, threads=2))
    (oa, ga, sa), (ob, gb, sb) = outs
    pix = py * W + px
    m2 = sa.field("means2D")[j]
    if not (abs(m2[0] - px) < 1e-4 and abs(m2[1] - py) < 1e-4):
        pytest.skip("projection did not land on the pixel centre: %s" % (m2,))
    assert sa.field("tie_pixels")[pix] == 1 and sa.field("tie_gaussians")[j] == 1
    moved = 0
    for n in ("color", "depth", "opacity", "semantic"):
        d = np.abs(oa[n].astype(np.float64) - ob[n]).reshape(oa[n].shape[0], -1)
        b = sa.img_bound(n).astype(np.float64)
        assert (d <= harness.TIE_SLACK * b[None, :] + 1e-6).all(), (n, float((d - b[None, :]).max()))
        if d[:, pix].max() > 1e-5:
            moved += 1
            assert b[pix] <= 3.0 * d[:, pix].max() + 1e-6, (n, b[pix], d[:, pix].max())   # the bound is not vacuous
    assert moved >= 3, "the flip did not happen"
    for n in ga:
        x, y = np.asarray(ga[n], np.float64), np.asarray(gb[n], np.float64).reshape(np.asarray(ga[n]).shape)
        if x.size == 0:
            continue
        b = np.asarray(sa.grad_bounds[n], np.float64).reshape(x.shape)
        mx = float(np.abs(x).max())
        d = np.abs(x - y)
        assert (d <= harness.TIE_SLACK * b + 2e-6 * mx).all(), (n, float((d - harness.TIE_SLACK * b).max() / mx))
    # rows of splats that never touch the flagged pixel carry no bound at all
    assert float(sa.grad_bounds["means3D"][~sa.field("tie_gaussians").astype(bool)].max(initial=0.0)) == 0.0
    sa.free(); sb.free()


def test_fp32_model_of_the_exponents_argument_rounding():
    """The noise-floor model of tests/harness.py (FP32_MODEL_ARG_ROUNDINGS; oracle/hsr_oracle.c hsro_set_exp_argument_error): the exponent's
    argument is a sum of three products that cancel on elongated, rotated splats, so another correct fp32 evaluation order lands G further
    away than one ulp of exp does — measured here on the oracle alone: the distance of the fp32 model from the truth build with the
    argument rounding on, against the same seeds with it off.  Off = the model as it was (bit for bit); on = a bounded, larger spread."""
    import harness
    W, H, P, K = 120, 72, 800, 4
    cam, sc, up = scenes.build(W, H, P, K, seed=77, kind="aniso", scale_mult=3.0)
    # needles: one axis 8x longer still
    sc["scales"] = sc["scales"].clone()
    sc["scales"][:, 2] = sc["scales"][:, 2] * 8.0
    kw = dict(harness.variant_kwargs(sc, "sr", None), semantics_precomp=sc["semantics_precomp"])
    g_ = {n: v.numpy() for n, v in up.items()}
    _, gt, st_t = run_oracle(cam, sc, up, semantic=True, precision="f64", bounds=False)
    st_t.free()
    _, _, st = run_oracle(cam, sc, up, semantic=True, bounds=False)
    def model(seed, e, a):
        return O.backward(st, cam, sc["means3D"], g_, median_rule="forward", fp32_atomics_seed=seed, exp_ulps=e, arg_roundings=a, **kw)
    def dist(g, n):
        t = np.asarray(gt[n], np.float64)
        return float(np.abs(np.asarray(g[n], np.float64).reshape(t.shape) - t).max() / np.abs(t).max())
    off = [model(s, 1.0, 0.0) for s in range(4)]
    on = [model(s, 1.0, harness.FP32_MODEL_ARG_ROUNDINGS) for s in range(4)]
    again = model(0, 1.0, 0.0)
    for n in ("means3D", "rotations", "scales", "opacities"):
        assert np.array_equal(off[0][n], again[n]), n          # seeded: the same model run twice is the same bits
        d_off, d_on = max(dist(g, n) for g in off), max(dist(g, n) for g in on)
        print("%-10s fp32 model vs truth, tensor-wide: orders + 1-ulp exp %.2e ; + argument rounding %.2e" % (n, d_off, d_on))
        assert d_on < 20.0 * max(d_off, 1e-5), (n, d_on, d_off)   # bounded: a rounding model, not a different function
    # dL_dopacity is a plain sum of G dL_dalpha (no ill-conditioned chain behind it): there the argument rounding shows by itself
    assert max(dist(g, "opacities") for g in on) > 2.0 * max(dist(g, "opacities") for g in off)
    st.free()
