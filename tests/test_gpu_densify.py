"""GPU parity for the densification step (include/hsr_densify.h) against oracle/densify_oracle.py, and the reference-named
add_new_gaussians_semantic on a small map."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,seed", [(48, 64, 0), (37, 53, 1), (680, 1200, 2), (16, 16, 3)])
def test_non_presence_points_match_oracle(H, W, seed):
    import densify_oracle as DO
    from hsr_utils import densify as D
    from test_densify import make_frame
    sil, rd, gt, col, K, w2c = make_frame(H, W, seed)
    t = lambda a: torch.tensor(a, device="cuda")
    pt_cld, msd, mask, ls = D.non_presence_points(t(sil), t(rd), t(gt), t(col), torch.tensor(K), t(w2c), 0.5)
    c2w = torch.inverse(torch.tensor(w2c)).numpy()
    o = DO.non_presence_points(sil, rd, gt, col, K, c2w, 0.5)
    assert np.array_equal(mask.cpu().numpy(), o["mask"])                      # integer decisions: bit-exact (incl. the median)
    assert pt_cld.shape[0] == int(o["mask"].sum())
    np.testing.assert_allclose(pt_cld[:, :3].cpu().numpy(), o["means3D"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(pt_cld[:, 3:].cpu().numpy(), o["rgb"])      # order-preserving compaction
    np.testing.assert_allclose(msd.cpu().numpy(), o["mean3_sq_dist"], rtol=1e-6)
    np.testing.assert_allclose(ls[:, 0].cpu().numpy(), o["log_scales"], rtol=1e-5, atol=1e-6)


def test_no_new_points_and_bad_input():
    from hsr_utils import densify as D
    from test_densify import make_frame
    sil, rd, gt, col, K, w2c = make_frame(32, 40, 5, hole=False)
    t = lambda a: torch.tensor(a, device="cuda")
    pt_cld, msd, mask, ls = D.non_presence_points(t(sil), t(gt), t(gt), t(col), torch.tensor(K), t(w2c), 0.5)
    assert pt_cld.shape == (0, 6) and not mask.any()
    with pytest.raises(RuntimeError, match="no CPU path"):
        D.non_presence_points(torch.tensor(sil), t(gt), t(gt), t(col), torch.tensor(K), t(w2c), 0.5)


def test_add_new_gaussians_semantic_grows_the_map_where_nothing_was_rendered():
    from hsr_utils import densify as D, setup_camera, make_scene
    W, H, K, P = 160, 120, 6, 3000
    kmat = np.array([[150.0, 0, 79.5], [0, 150.0, 59.5], [0, 0, 1]])
    cam = setup_camera(W, H, kmat, np.eye(4), device="cuda")
    sc = make_scene(P, W, H, K, kmat, seed=4, scale_mult=3.0)
    keep = sc["means3D"][:, 0] < 0                                            # a map that only covers the left half
    params = {"means3D": sc["means3D"][keep], "unnorm_rotations": sc["rotations"][keep],
              "logit_opacities": torch.full((int(keep.sum()), 1), 4.0), "log_scales": sc["scales"][keep][:, :1].log(),
              "rgb_colors": sc["colors_precomp"][keep], "semantic": sc["semantics_precomp"][keep]}
    params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in params.items()}
    params["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0]).view(1, 4, 1).repeat(1, 1, 2).cuda())
    params["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 2).cuda())
    n0 = params["means3D"].shape[0]
    variables = {"timestep": torch.zeros(n0, device="cuda")}
    curr = {"cam": cam, "w2c": torch.eye(4, device="cuda"), "depth": torch.full((1, H, W), 2.0, device="cuda"),
            "im": torch.rand(3, H, W, device="cuda"), "intrinsics": torch.tensor(kmat, dtype=torch.float32)}
    params, variables = D.add_new_gaussians_semantic(params, variables, curr, 0.5, 1, "projective", K)
    n1 = params["means3D"].shape[0]
    added = params["means3D"][n0:]
    assert n1 > n0 and variables["timestep"].shape[0] == n1 and (variables["timestep"][n0:] == 1).all()
    assert float((added[:, 0] > -0.1).float().mean()) > 0.9                   # new points fill the uncovered right half
    np.testing.assert_allclose(added[:, 2].detach().cpu().numpy(), 2.0, rtol=1e-5)   # at the measured depth
    for k in ("rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales", "semantic"):
        assert params[k].shape[0] == n1 and isinstance(params[k], torch.nn.Parameter) and params[k].requires_grad
    assert variables["means2D_gradient_accum"].shape == (n1,) and variables["denom"].shape == (n1,)
