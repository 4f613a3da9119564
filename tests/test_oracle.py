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


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(GOLD) if f.endswith(".npz") and not f.startswith("loss_")) if os.path.isdir(GOLD) else [])
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
