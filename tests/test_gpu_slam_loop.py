"""End-to-end use of the three rows as scripts/hierslam.py chains them: a tracking loop (pose only, :1683-1860) and a few
mapping iterations (Gaussians, :2016-2040) on a synthetic scene, with torch.optim.Adam as in the reference.  Checks what
only correct gradients through the whole chain can deliver: the camera pose converges back to the ground truth, and the
mapping loss goes down."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(P=20000, W=320, H=240, K=12):
    from hsr_utils import setup_camera, make_scene
    kmat = np.array([[300.0, 0, 159.5], [0, 300.0, 119.5], [0, 0, 1]])
    cam = setup_camera(W, H, kmat, np.eye(4), device="cuda")
    sc = make_scene(P, W, H, K, kmat, seed=21, scale_mult=2.0)
    params = {"means3D": sc["means3D"], "unnorm_rotations": sc["rotations"],
              "logit_opacities": torch.logit(sc["opacities"].clamp(0.02, 0.98)), "log_scales": sc["scales"][:, :1].log(),
              "rgb_colors": sc["colors_precomp"], "semantic": sc["semantics_precomp"]}
    return cam, {k: v.clone().cuda() for k, v in params.items()}, (W, H, K)


def _render(params, cam, time_idx, gaussians_grad, camera_grad):
    from diff_gaussian_rasterization import GaussianRasterizer_semantic
    from hsr_utils import slam_helpers as SH
    rv = SH.transformed_params2rendervar_semantic(params, SH.transform_to_frame(params, time_idx, gaussians_grad, camera_grad))
    return GaussianRasterizer_semantic(raster_settings=cam)(**rv)


def test_tracking_recovers_the_camera_pose():
    from hsr_utils import losses as L
    cam, params, (W, H, K) = _scene()
    gt_q = torch.tensor([0.9990, 0.020, -0.030, 0.025]); gt_q = gt_q / gt_q.norm()
    gt_t = torch.tensor([0.030, -0.020, 0.040])
    rots = torch.zeros(1, 4, 2); rots[0, 0, :] = 1.0; rots[0, :, 1] = gt_q
    trans = torch.zeros(1, 3, 2); trans[0, :, 1] = gt_t
    params["cam_unnorm_rots"], params["cam_trans"] = rots.cuda(), trans.cuda()
    with torch.no_grad():
        im_gt, _, _, depth_gt, _, _ = _render(params, cam, 1, False, False)
    # start tracking frame 1 from the pose of frame 0 (identity), as the reference initialises a new frame (:1622-1650)
    params["cam_unnorm_rots"] = rots.clone().cuda(); params["cam_unnorm_rots"][0, :, 1] = torch.tensor([1.0, 0, 0, 0])
    params["cam_trans"] = torch.zeros(1, 3, 2).cuda()
    params["cam_unnorm_rots"].requires_grad_(True); params["cam_trans"].requires_grad_(True)
    opt = torch.optim.Adam([{"params": [params["cam_unnorm_rots"]], "lr": 4e-4}, {"params": [params["cam_trans"]], "lr": 2e-3}])

    def pose_err():
        q = torch.nn.functional.normalize(params["cam_unnorm_rots"][0, :, 1].detach().cpu(), dim=0)
        return float(1 - abs(float((q * gt_q).sum()))), float((params["cam_trans"][0, :, 1].detach().cpu() - gt_t).norm())
    r0, t0 = pose_err()
    first = last = None
    for it in range(200):
        im, radius, sem, depth, med, opac = _render(params, cam, 1, False, True)
        mask = ((depth_gt > 0) & (opac > 0.5)).detach()
        loss = L.masked_l1(depth, depth_gt, mask, "sum") + 0.5 * L.masked_l1(im, im_gt, mask, "sum")   # hierslam.py:925, :935
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        first = float(loss.detach()) if first is None else first
        last = float(loss.detach())
    r1, t1 = pose_err()
    assert last < 0.2 * first, (first, last)
    assert t1 < 0.2 * t0 and r1 < 0.2 * r0, ((r0, t0), (r1, t1))
    # the other frame's pose column received no gradient
    assert torch.equal(params["cam_trans"][0, :, 0].detach().cpu(), torch.zeros(3))


def test_mapping_iterations_reduce_the_loss():
    from hsr_utils import losses as L
    cam, params, (W, H, K) = _scene()
    params["cam_unnorm_rots"] = torch.tensor([1.0, 0, 0, 0]).view(1, 4, 1).cuda()
    params["cam_trans"] = torch.zeros(1, 3, 1).cuda()
    with torch.no_grad():
        im_gt, _, sem_gt, depth_gt, _, _ = _render(params, cam, 0, False, False)
    sizes = [4, 8]
    lab = torch.stack([sem_gt[:4].argmax(dim=0), sem_gt[4:12].argmax(dim=0)])
    g = torch.Generator().manual_seed(3)
    for k in ("means3D", "rgb_colors", "semantic", "logit_opacities", "log_scales"):   # disturb the map, then re-fit it
        params[k] = (params[k] + 0.05 * torch.randn(params[k].shape, generator=g).cuda() * params[k].abs().mean()).requires_grad_(True)
    mlp = torch.nn.Conv2d(K, 20, kernel_size=1).cuda()
    leaf_lab = torch.randint(0, 20, (H, W), generator=g).cuda()
    lrs = {"means3D": 1e-4, "rgb_colors": 2.5e-3, "semantic": 2.5e-3, "logit_opacities": 0.05, "log_scales": 1e-3}   # configs/replica
    opt = torch.optim.Adam([{"params": [params[k]], "lr": lr} for k, lr in lrs.items()] + [{"params": mlp.parameters(), "lr": 1e-3}])
    hist = []
    for it in range(40):
        im, radius, sem, depth, med, opac = _render(params, cam, 0, True, False)
        mask = (depth_gt > 0).detach()
        loss = 0.5 * L.mapping_image_loss(im, im_gt) + L.masked_l1(depth, depth_gt, mask, "mean") \
            + 0.01 * (L.tree_cross_entropy(sem, lab, sizes) + 5.0 * L.leaf_mlp_cross_entropy(sem, mlp, leaf_lab))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        hist.append(float(loss.detach()))
    assert all(np.isfinite(hist)) and hist[-1] < 0.8 * hist[0], (hist[0], hist[-1])


def test_mapping_with_map_surgery_in_the_loop():
    """mapping iterations with the map changing size under the optimizer, as the reference's loop does between frames: opacity prune
    (utils/slam_external.py:167-188) and gradient-driven densification (:191-242) every few iterations, Adam moments carried through
    the fused compaction, the non-blocking forward switched on (a new P is a new size: that call falls back to the blocking path).
    The loss keeps going down across the surgery and every gradient stays finite."""
    import diff_gaussian_rasterization as dgr
    from hsr_utils import losses as L, slam_external as SE
    cam, params, (W, H, K) = _scene(P=12000)
    params["cam_unnorm_rots"] = torch.tensor([1.0, 0, 0, 0]).view(1, 4, 1).cuda()
    params["cam_trans"] = torch.zeros(1, 3, 1).cuda()
    with torch.no_grad():
        im_gt, _, sem_gt, depth_gt, _, _ = _render(params, cam, 0, False, False)
    sizes = [4, 8]
    lab = torch.stack([sem_gt[:4].argmax(dim=0), sem_gt[4:12].argmax(dim=0)])
    g = torch.Generator().manual_seed(5)
    for k in ("means3D", "rgb_colors", "semantic", "logit_opacities", "log_scales"):
        params[k] = (params[k] + 0.05 * torch.randn(params[k].shape, generator=g).cuda() * params[k].abs().mean())
    params = {k: torch.nn.Parameter(v.detach().clone().requires_grad_(True)) for k, v in params.items()}
    lrs = {"means3D": 1e-4, "rgb_colors": 2.5e-3, "unnorm_rotations": 1e-3, "semantic": 2.5e-3, "logit_opacities": 0.05, "log_scales": 1e-3,
           "cam_unnorm_rots": 0.0, "cam_trans": 0.0}
    opt = torch.optim.Adam([{"params": [params[k]], "name": k, "lr": lrs[k]} for k in params])
    P0 = params["means3D"].shape[0]
    variables = {"means2D_gradient_accum": torch.zeros(P0).cuda(), "denom": torch.zeros(P0).cuda(), "max_2D_radius": torch.zeros(P0).cuda(),
                 "timestep": torch.zeros(P0).cuda(), "scene_radius": torch.tensor(float(depth_gt.max()) / 3.0).cuda()}
    dd = dict(start_after=5, remove_big_after=10 ** 9, stop_after=10 ** 9, densify_every=10, grad_thresh=2e-5, num_to_split_into=2,
              removal_opacity_threshold=0.03, final_removal_opacity_threshold=0.03, reset_opacities=False, reset_opacities_every=10 ** 9)
    pd = dict(start_after=0, remove_big_after=10 ** 9, stop_after=10 ** 9, prune_every=7, removal_opacity_threshold=0.03,
              final_removal_opacity_threshold=0.03, reset_opacities=False, reset_opacities_every=10 ** 9)
    prev = dgr.set_async_forward(True)
    hist, sizes_seen, changed_at = [], set(), []
    try:
        for it in range(45):
            from diff_gaussian_rasterization import GaussianRasterizer_semantic
            from hsr_utils import slam_helpers as SH
            rv = SH.transformed_params2rendervar_semantic(params, SH.transform_to_frame(params, 0, True, False))
            rv["means2D"].retain_grad()                                     # scripts/hierslam.py:895
            im, radius, sem, depth, med, opac = GaussianRasterizer_semantic(raster_settings=cam)(**rv)
            mask = (depth_gt > 0).detach()
            loss = 0.5 * L.mapping_image_loss(im, im_gt) + L.masked_l1(depth, depth_gt, mask, "mean") \
                + 0.01 * L.tree_cross_entropy(sem, lab, sizes)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            for k in ("means3D", "semantic", "log_scales", "unnorm_rotations"):
                assert torch.isfinite(params[k].grad).all(), (it, k)
            opt.step()
            hist.append(float(loss.detach()))
            variables["means2D"], variables["seen"] = rv["means2D"], radius > 0    # :1102-1104
            with torch.no_grad():
                params, variables = SE.prune_gaussians(params, variables, opt, it, pd)
            if params["means3D"].shape[0] == variables["seen"].shape[0]:    # (a prune this iteration changed P: `seen` is per old row)
                params, variables = SE.densify(params, variables, opt, it, dd)
            if sizes_seen and int(params["means3D"].shape[0]) not in sizes_seen:
                changed_at.append(it)
            sizes_seen.add(int(params["means3D"].shape[0]))
    finally:
        dgr.set_async_forward(prev)
    assert len(sizes_seen) >= 3, sizes_seen                     # the map really shrank and grew
    # cloning and splitting change the image (a clone doubles its Gaussian's density, as in the reference): the loss may jump at a
    # surgery; between surgeries the carried-over Adam state keeps it going down
    assert all(np.isfinite(hist))
    bounds = [0] + [c + 1 for c in changed_at] + [len(hist)]
    checked = 0
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b - a >= 4:
            assert hist[b - 1] < hist[a], (a, b, hist[a], hist[b - 1])
            checked += 1
    assert checked >= 2, (changed_at, len(hist))
    for k, v in params.items():                                  # the optimizer still owns exactly the live parameters
        assert [gr for gr in opt.param_groups if gr["name"] == k][0]["params"][0] is v
