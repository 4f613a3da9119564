"""GPU parity for the fused rasterizer-input preparation (include/hsr_frame_prep.h) against oracle/frame_prep_oracle.py,
against torch eager on the same device, and chained into the rasterizer."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu

FWD_RTOL, FWD_ATOL = 2e-6, 2e-6   # fp32 forward: same expression tree as the oracle, libm-vs-device exp/sqrt ulps
BWD_TOL = 1e-4                    # gradients vs the float64 oracle, relative to the largest entry of each tensor


def _params(P, S, frames=6, seed=0, dev="cuda"):
    from test_frame_prep import make_inputs
    inp = make_inputs(P, S, frames, seed)
    g = np.random.default_rng(seed + 100)
    t = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in inp.items()}
    t["rgb_colors"] = torch.tensor(g.random((P, 3)).astype(np.float32), device=dev, requires_grad=True)
    t["semantic"] = torch.tensor(g.random((P, 5)).astype(np.float32), device=dev, requires_grad=True)
    return inp, t


def _close(name, got, want, tol):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = max(np.abs(want).max() if want.size else 0.0, 1e-30)
    err = np.abs(got - want).max() / scale if want.size else 0.0
    assert err <= tol, "%s: max err %.3g of scale %.3g" % (name, err, scale)


VARIANTS = [("semantic", 1), ("rendervar", 1), ("silhouette", 1), ("depthsil", 1), ("rendervar", 3), ("depthsil", 3), ("semantic", 3)]


@pytest.mark.parametrize("variant,S", VARIANTS)
@pytest.mark.parametrize("P", [1, 1000, 70001])
def test_prep_matches_oracle(variant, S, P):
    import frame_prep_oracle as O
    from test_frame_prep import make_grads
    from hsr_utils import slam_helpers as SH
    tidx = 4
    inp, t = _params(P, S)
    w2c_np = None
    if variant == "depthsil":
        w2c_np = np.eye(4, dtype=np.float32); w2c_np[:3, :3] = O._rotation(np.array([0.9, 0.1, -0.3, 0.2], np.float32))[0]; w2c_np[:3, 3] = (0.1, -0.2, 0.3)
    w2c = None if w2c_np is None else torch.tensor(w2c_np, device="cuda")
    tg = SH.transform_to_frame(t, tidx, gaussians_grad=True, camera_grad=True)
    if variant == "semantic":
        rv = SH.transformed_params2rendervar_semantic(t, tg); rot_source = O.ROT_PARAMS
    elif variant == "rendervar":
        rv = SH.transformed_params2rendervar(t, tg); rot_source = O.ROT_TRANSFORMED
    elif variant == "silhouette":
        rv = SH.transformed_params2silhouette(t, tg); rot_source = O.ROT_TRANSFORMED
    else:
        rv = SH.transformed_params2depthplussilhouette(t, w2c, tg); rot_source = O.ROT_TRANSFORMED
    fo = O.forward(**inp, time_idx=tidx, rot_source=rot_source, w2c=w2c_np)
    for k_rv, k_o in (("means3D", "means3D"), ("rotations", "rotations"), ("opacities", "opacities"), ("scales", "scales")):
        np.testing.assert_allclose(rv[k_rv].detach().cpu().numpy(), fo[k_o], rtol=FWD_RTOL, atol=FWD_ATOL, err_msg=k_rv)
    np.testing.assert_allclose(tg["unnorm_rotations"].detach().cpu().numpy(), fo["unnorm_rotations"], rtol=FWD_RTOL, atol=FWD_ATOL)
    assert tg["means3D"] is rv["means3D"]
    assert rv["means2D"].shape == (P, 3) and rv["means2D"].requires_grad and not rv["means2D"].detach().any()
    if variant == "semantic":
        assert rv["semantics_precomp"] is t["semantic"] and rv["colors_precomp"] is t["rgb_colors"]
    if variant == "silhouette":
        assert (rv["colors_precomp"][:, 0] == 1).all() and not rv["colors_precomp"][:, 1:].any()
    if variant == "depthsil":
        np.testing.assert_allclose(rv["colors_precomp"].detach().cpu().numpy(), fo["depth_sil"], rtol=1e-5, atol=1e-5)
    grads = make_grads(P, variant == "depthsil")
    gt = {k: torch.tensor(v, device="cuda") for k, v in grads.items()}
    loss = (rv["means3D"] * gt["means3D"]).sum() + (tg["unnorm_rotations"] * gt["unnorm_rotations"]).sum() + \
        (rv["rotations"] * gt["rotations"]).sum() + (rv["opacities"] * gt["opacities"]).sum() + (rv["scales"] * gt["scales"]).sum()
    if variant == "depthsil":
        loss = loss + (rv["colors_precomp"] * gt["depth_sil"]).sum()
    loss.backward()
    bo = O.backward(**inp, time_idx=tidx, grads=grads, rot_source=rot_source, w2c=w2c_np)
    for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales"):
        _close(k, t[k].grad.cpu().numpy(), bo[k], BWD_TOL)
    _close("cam_unnorm_rot", t["cam_unnorm_rots"].grad[0, :, tidx].cpu().numpy(), bo["cam_unnorm_rot"], BWD_TOL)
    _close("cam_tran", t["cam_trans"].grad[0, :, tidx].cpu().numpy(), bo["cam_tran"], BWD_TOL)
    other = t["cam_unnorm_rots"].grad.clone(); other[0, :, tidx] = 0
    assert not other.any()


@pytest.mark.parametrize("gaussians_grad,camera_grad", [(True, False), (False, True), (False, False)])
@pytest.mark.parametrize("variant", ["semantic", "rendervar"])
def test_detach_flags_follow_the_reference(gaussians_grad, camera_grad, variant):
    """slam_helpers.py:292-314: camera_grad=False detaches the pose; gaussians_grad=False detaches means3D and the
    transformed quaternions, while sigmoid / exp / F.normalize(params[...]) in the rendervar builders keep their gradient."""
    from hsr_utils import slam_helpers as SH
    _, t = _params(500, 1)
    tg = SH.transform_to_frame(t, 2, gaussians_grad=gaussians_grad, camera_grad=camera_grad)
    rv = (SH.transformed_params2rendervar_semantic if variant == "semantic" else SH.transformed_params2rendervar)(t, tg)
    (rv["means3D"].sum() + rv["rotations"].square().sum() * 0 + (rv["rotations"] * torch.arange(4, device="cuda")).sum()
     + rv["opacities"].sum() + rv["scales"].sum()).backward()
    assert (t["means3D"].grad is not None) == gaussians_grad
    assert (t["cam_unnorm_rots"].grad is not None) == camera_grad and (t["cam_trans"].grad is not None) == camera_grad
    assert t["logit_opacities"].grad is not None and t["log_scales"].grad is not None
    rot_has_grad = t["unnorm_rotations"].grad is not None
    assert rot_has_grad == (gaussians_grad or variant == "semantic")


def test_prep_is_reproducible_and_handles_empty():
    from hsr_utils import slam_helpers as SH
    outs = []
    for _ in range(2):
        _, t = _params(50000, 3, seed=3)
        tg = SH.transform_to_frame(t, 1, True, True)
        rv = SH.transformed_params2rendervar(t, tg)
        (rv["means3D"].square().sum() + rv["rotations"][:, 1].sum()).backward()
        outs.append((t["cam_unnorm_rots"].grad.clone(), t["cam_trans"].grad.clone(), t["means3D"].grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    _, t = _params(0, 1)
    tg = SH.transform_to_frame(t, 0, True, True)
    rv = SH.transformed_params2rendervar_semantic(t, tg)
    assert rv["means3D"].shape == (0, 3) and rv["scales"].shape == (0, 3)
    (rv["means3D"].sum() + rv["opacities"].sum()).backward()
    assert not t["cam_trans"].grad.any()


def test_prep_rejects_bad_input():
    from hsr_utils import slam_helpers as SH
    _, t = _params(10, 1)
    with pytest.raises(RuntimeError):
        SH.transform_to_frame(t, 99, True, True)["means3D"]
    cpu = {k: v.detach().cpu() for k, v in t.items()}
    with pytest.raises(RuntimeError):
        SH.transform_to_frame(cpu, 0, True, True)["means3D"]
    with pytest.raises(TypeError):
        SH.transformed_params2rendervar(t, {"means3D": t["means3D"], "unnorm_rotations": t["unnorm_rotations"]})


def _eager_prep_semantic(p, tidx):
    """torch eager, same device: the op chain the reference runs (slam_helpers.py:278-330, :195-219), written with
    device-agnostic calls; used as the plain-PyTorch comparator for the chained test."""
    import torch.nn.functional as F
    q = F.normalize(p['cam_unnorm_rots'][..., tidx])
    n = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = n[0]
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)]),
                     torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)]),
                     torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)])])
    means = p['means3D'] @ R.T + p['cam_trans'][0, :, tidx]
    return {'means3D': means, 'colors_precomp': p['rgb_colors'], 'rotations': F.normalize(p['unnorm_rotations']),
            'opacities': torch.sigmoid(p['logit_opacities']), 'scales': torch.exp(torch.tile(p['log_scales'], (1, 3))),
            'semantics_precomp': p['semantic'], 'means2D': torch.zeros_like(p['means3D'], requires_grad=True) + 0}


def test_prep_chained_into_rasterizer_matches_eager_chain():
    """tracking-style step: pose + Gaussians -> fused prep -> semantic rasterizer -> loss; all parameter gradients against
    the same chain with torch eager ops in place of the fused kernel."""
    from diff_gaussian_rasterization import GaussianRasterizer_semantic
    from hsr_utils import slam_helpers as SH, setup_camera, make_scene
    W, H, K, P = 320, 240, 5, 20000
    kmat = np.array([[300.0, 0, 159.5], [0, 300.0, 119.5], [0, 0, 1]])
    cam = setup_camera(W, H, kmat, np.eye(4), device="cuda")
    sc = make_scene(P, W, H, K, kmat, seed=5)
    g = torch.Generator().manual_seed(0)

    def fresh():
        p = {"means3D": sc["means3D"], "unnorm_rotations": sc["rotations"] * 1.3, "logit_opacities": torch.logit(sc["opacities"].clamp(1e-4, 1 - 1e-4)),
             "log_scales": sc["scales"][:, :1].log(), "rgb_colors": sc["colors_precomp"], "semantic": sc["semantics_precomp"]}
        p = {k: v.clone().cuda().requires_grad_(True) for k, v in p.items()}
        rots = torch.zeros(1, 4, 3); rots[0, 0] = 1.0; rots[0, :, 1] = torch.tensor([0.999, 0.01, -0.02, 0.015]) * 1.1
        trans = torch.zeros(1, 3, 3); trans[0, :, 1] = torch.tensor([0.02, -0.01, 0.03])
        p["cam_unnorm_rots"] = rots.cuda().requires_grad_(True); p["cam_trans"] = trans.cuda().requires_grad_(True)
        return p
    wts = [torch.randn(c, H, W, generator=g).cuda() / (W * H) for c in (3, K, 1, 1, 1)]

    def step(p, rv):
        rv['means2D'].retain_grad()
        im, radius, sem, depth, med, opac = GaussianRasterizer_semantic(raster_settings=cam)(**rv)
        loss = (im * wts[0]).sum() + (sem * wts[1]).sum() + (depth * wts[2]).sum() + (med * wts[3]).sum() + (opac * wts[4]).sum()
        loss.backward()
        return {k: v.grad.clone() for k, v in p.items()}, rv['means2D'].grad.clone(), im.detach()
    pa = fresh()
    rva = SH.transformed_params2rendervar_semantic(pa, SH.transform_to_frame(pa, 1, True, True))
    ga, m2a, ima = step(pa, rva)
    pb = fresh()
    rvb = _eager_prep_semantic(pb, 1)
    # a last-ulp difference in a mean can flip a depth order or a 1/255 test in the rasterizer, which is not what this test
    # is about: give the eager chain the fused kernel's VALUES (its autograd graph stays the eager one)
    for k in ("means3D", "rotations", "opacities", "scales"):
        _close("eager vs fused " + k, rva[k].detach().cpu().numpy(), rvb[k].detach().cpu().numpy(), 1e-5)
        rvb[k].data.copy_(rva[k].data)
    gb, m2b, imb = step(pb, rvb)
    assert torch.equal(ima, imb)
    _close("means2D.grad", m2a.cpu().numpy(), m2b.cpu().numpy(), 1e-5)
    for k in ga:
        _close(k + ".grad", ga[k].cpu().numpy(), gb[k].cpu().numpy(), 1e-4)


def test_depth_silhouette_bundle_is_not_reused_for_another_w2c():
    """a TransformedGaussians caches the bundle of its first launch; a second depth+silhouette request with a
    DIFFERENT world-to-camera matrix (or the same tensor changed in place) must be recomputed, not served from the cache."""
    from hsr_utils import slam_helpers as SH
    inp, t = _params(500, 1)
    tg = SH.transform_to_frame(t, 2, gaussians_grad=False, camera_grad=False)
    w_a = torch.eye(4, device="cuda")
    w_b = torch.eye(4, device="cuda"); w_b[2, 3] = 0.75
    rv_a = SH.transformed_params2depthplussilhouette(t, w_a, tg)
    rv_a2 = SH.transformed_params2depthplussilhouette(t, w_a, tg)
    assert rv_a2["colors_precomp"] is rv_a["colors_precomp"]                  # same matrix: the cached launch serves it
    rv_b = SH.transformed_params2depthplussilhouette(t, w_b, tg)
    d_a, d_b = rv_a["colors_precomp"][:, 0], rv_b["colors_precomp"][:, 0]
    assert torch.allclose(d_b, d_a + 0.75, atol=1e-5)                          # depth moved with the camera
    w_a[2, 3] = -0.25                                                          # in-place edit of the first matrix
    rv_c = SH.transformed_params2depthplussilhouette(t, w_a, tg)
    assert torch.allclose(rv_c["colors_precomp"][:, 0], d_a - 0.25, atol=1e-5)
