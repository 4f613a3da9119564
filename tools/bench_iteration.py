#!/usr/bin/env python3
"""One mapping iteration of the SLAM loop around the rasterizer (scripts/hierslam.py:870-1016, :2016-2037 without the
optimizer): world->camera preparation, semantic render, mapping losses, backward to every Gaussian parameter and the pose.
Times (a) the fused preparation + rasterizer + fused loss heads of this repo and (b) the same rasterizer with the
reference's torch eager chains either side of it.  One JSON line.

    python tools/bench_iteration.py [--P 500000] [--iters 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hier-slam_amd"))


REPEATS = 3


def measure(P=500000, W=1200, H=680, iters=20):
    from diff_gaussian_rasterization import GaussianRasterizer_semantic
    from hsr_utils import slam_helpers as SH, losses as L, setup_camera, make_scene
    from hsr_utils.camera import replica_intrinsics
    sizes = [2, 4, 6, 6, 8]
    K = sum(sizes)
    kmat = replica_intrinsics(W, H)
    cam = setup_camera(W, H, kmat, np.eye(4), device="cuda")
    sc = make_scene(P, W, H, K, kmat, seed=0)
    g = torch.Generator().manual_seed(0)
    params = {"means3D": sc["means3D"], "unnorm_rotations": sc["rotations"], "logit_opacities": torch.logit(sc["opacities"].clamp(1e-4, 1 - 1e-4)),
              "log_scales": sc["scales"][:, :1].log(), "rgb_colors": sc["colors_precomp"], "semantic": sc["semantics_precomp"]}
    params = {k: v.clone().cuda().requires_grad_(True) for k, v in params.items()}
    rots = torch.zeros(1, 4, 4); rots[0, 0] = 1.0
    params["cam_unnorm_rots"] = rots.cuda().requires_grad_(True)
    params["cam_trans"] = torch.zeros(1, 3, 4).cuda().requires_grad_(True)
    gt_im, gt_d = torch.rand(3, H, W, generator=g).cuda(), (torch.rand(1, H, W, generator=g) * 5 + 0.5).cuda()
    lab = torch.stack([torch.randint(0, n, (H, W), generator=g) for n in sizes]).cuda()
    mlp = torch.nn.Conv2d(K, 102, kernel_size=1).cuda()          # MLP_func, scripts/hierslam.py:1756
    leaf_lab = torch.randint(0, 102, (H, W), generator=g).cuda()
    lab_with_leaf = torch.cat((lab, leaf_lab[None]), 0)
    w1 = torch.tensor([np.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    w1 = (w1 / w1.sum()).unsqueeze(1)
    win = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0).expand(3, 1, 11, 11).contiguous().cuda()

    def eager_prep(p, tidx):
        q = F.normalize(p['cam_unnorm_rots'][..., tidx])
        nq = q / torch.sqrt((q * q).sum(dim=1))[:, None]
        r, x, y, z = nq[:, 0], nq[:, 1], nq[:, 2], nq[:, 3]
        rot = torch.zeros((1, 3, 3), device="cuda")
        rot[:, 0, 0] = 1 - 2 * (y * y + z * z); rot[:, 0, 1] = 2 * (x * y - r * z); rot[:, 0, 2] = 2 * (x * z + r * y)
        rot[:, 1, 0] = 2 * (x * y + r * z); rot[:, 1, 1] = 1 - 2 * (x * x + z * z); rot[:, 1, 2] = 2 * (y * z - r * x)
        rot[:, 2, 0] = 2 * (x * z - r * y); rot[:, 2, 1] = 2 * (y * z + r * x); rot[:, 2, 2] = 1 - 2 * (x * x + y * y)
        rel = torch.eye(4, device="cuda"); rel[:3, :3] = rot[0]; rel[:3, 3] = p['cam_trans'][0, :, tidx]
        pts4 = torch.cat((p['means3D'], torch.ones(P, 1, device="cuda")), dim=1)
        return {'means3D': (rel @ pts4.T).T[:, :3], 'colors_precomp': p['rgb_colors'], 'rotations': F.normalize(p['unnorm_rotations']),
                'opacities': torch.sigmoid(p['logit_opacities']), 'scales': torch.exp(torch.tile(p['log_scales'], (1, 3))),
                'semantics_precomp': p['semantic'], 'means2D': torch.zeros_like(p['means3D'], requires_grad=True) + 0}

    def eager_ssim(a, b):
        mu1, mu2 = F.conv2d(a, win, padding=5, groups=3), F.conv2d(b, win, padding=5, groups=3)
        s1 = F.conv2d(a * a, win, padding=5, groups=3) - mu1.pow(2)
        s2 = F.conv2d(b * b, win, padding=5, groups=3) - mu2.pow(2)
        s12 = F.conv2d(a * b, win, padding=5, groups=3) - mu1 * mu2
        return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1.pow(2) + mu2.pow(2) + 1e-4) * (s1 + s2 + 9e-4))).mean()

    def iteration(fused, leaf=False):
        rv = SH.transformed_params2rendervar_semantic(params, SH.transform_to_frame(params, 1, True, False)) if fused else eager_prep(params, 1)
        rv['means2D'].retain_grad()
        im, radius, sem, depth, med, opac = GaussianRasterizer_semantic(raster_settings=cam)(**rv)
        if not fused:
            mask = ((gt_d > 0) & ~torch.isnan(depth)).detach()
        if fused:
            # the weighted dictionary sum of scripts/hierslam.py:1003-1016 as one node (L.weighted_sum) instead of Python arithmetic on 0-dim tensors
            terms = [L.l1_loss_v1(im, gt_im), L.calc_ssim(im, gt_im), L.mapping_depth_loss(depth, gt_d)]
            weights = [0.5 * 0.8, -0.5 * 0.2, 1.0, 1.0]
            if leaf:   # losses['sem'] with the leaf head on, one node (scripts/hierslam.py:963-983)
                terms.append(L.semantic_loss_mlp(sem, lab_with_leaf, sizes, mlp, weight_sem=(0.1, 0.5)))
            else:
                terms.append(L.tree_cross_entropy(sem, lab, sizes, weights=[0.1] * len(sizes)))
            loss = L.weighted_sum(terms, weights, constant=0.5 * 0.2)
        else:
            ce, b = 0.0, 0
            celoss = torch.nn.CrossEntropyLoss()
            for i, n in enumerate(sizes):
                lvl = sem[b:b + n].permute(1, 2, 0)
                ce = ce + celoss(lvl.reshape(-1, n), lab[i].view(-1).long())
                b += n
            loss = 0.5 * (0.8 * torch.abs(im - gt_im).mean() + 0.2 * (1.0 - eager_ssim(im, gt_im))) + torch.abs(gt_d - depth)[mask].mean() + 0.1 * ce
            if leaf:
                logits = mlp(sem.unsqueeze(0))
                logits = logits.squeeze(0).view(logits.shape[1], -1).permute(1, 0)
                loss = loss + 0.5 * celoss(logits, leaf_lab.view(-1).long())
        loss.backward()
        return loss

    def timeit(fused, leaf=False):
        for _ in range(3):
            iteration(fused, leaf)
        best = float("inf")
        for _ in range(REPEATS):   # host-side times: the fastest of a few repeats (a box's other tenants show up as +10-20 %)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                for v in list(params.values()) + list(mlp.parameters()):
                    v.grad = None
                iteration(fused, leaf)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / iters * 1e3)
        return best
    def tracking(fused, pose_only=False):
        """scripts/hierslam.py:1683-1860 per iteration: pose gradients only, masked L1 sums on depth and colour.  pose_only: the map
        tensors are detached (the reference leaves them attached with learning rate 0, so their gradients are computed and thrown
        away); the rasterizer then runs its geometry-only backward"""
        rv = SH.transformed_params2rendervar_semantic(params, SH.transform_to_frame(params, 1, False, True)) if fused else eager_prep(params, 1)
        if pose_only:
            rv = {k: (v if k in ("means3D", "means2D") or v is None else v.detach()) for k, v in rv.items()}
        im, radius, sem, depth, med, opac = GaussianRasterizer_semantic(raster_settings=cam)(**rv)
        if fused and os.environ.get("HSR_ITER_PLAIN_SUM"):   # round 3's composition: torch mask + two masked L1 heads + Python arithmetic
            mask = ((gt_d > 0) & ~torch.isnan(depth) & (opac > 0.99)).detach()
            loss = L.masked_l1(depth, gt_d, mask, "sum") + 0.5 * L.masked_l1(im, gt_im, mask, "sum")
        elif fused:   # the tracking branch of get_loss* as one node (mask, both sums, the weights)
            loss = L.tracking_loss(im, gt_im, depth, gt_d, opac, sil_thres=0.99, loss_weights={"im": 0.5, "depth": 1.0})
        else:
            mask = ((gt_d > 0) & ~torch.isnan(depth) & (opac > 0.99)).detach()
            loss = torch.abs(gt_d - depth)[mask].sum() + 0.5 * torch.abs(gt_im - im)[torch.tile(mask, (3, 1, 1))].sum()
        loss.backward()
        return loss

    def time_tracking(fused, pose_only=False):
        for _ in range(3):
            tracking(fused, pose_only)
        best = float("inf")
        for _ in range(REPEATS):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                for v in params.values():
                    v.grad = None
                tracking(fused, pose_only)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / iters * 1e3)
        return best
    if os.environ.get("HSR_ITER_ONLY") == "mapping":     # for a kernel trace of the fused mapping iteration alone (tools/ktrace_iter.sh)
        return {"fused_ms": timeit(True)}
    if os.environ.get("HSR_ITER_ONLY") == "leaf":
        return {"with_leaf_head_fused_ms": timeit(True, True)}
    if os.environ.get("HSR_ITER_ONLY") == "tracking":
        return {"tracking_fused_map_detached_ms": time_tracking(True, True)}
    lf, le = float(iteration(True).detach()), float(iteration(False).detach())
    # the opt-in non-blocking forward (diff_gaussian_rasterization.set_async_forward): the host no longer waits for num_rendered in
    # every frame, so the Python side of the iteration overlaps the device side of the previous one
    from diff_gaussian_rasterization import set_async_forward
    was = set_async_forward(True)
    run_ahead = {"note": "same fused iterations with set_async_forward(True)", "fused_ms": timeit(True), "tracking_fused_ms": time_tracking(True),
                 "tracking_fused_map_detached_ms": time_tracking(True, True), "with_leaf_head_fused_ms": timeit(True, True),
                 "loss_fused": float(iteration(True).detach())}
    set_async_forward(was)
    return {"workload": "mapping iteration without optimizer: prep + semantic render + mapping losses + backward, %dx%d, P=%d, K=%d" % (W, H, P, K),
            "fused_ms": timeit(True), "eager_around_same_rasterizer_ms": timeit(False), "loss_fused": lf, "loss_eager": le,
            "tracking_iteration": {"note": "pose-only iteration (scripts/hierslam.py:1683-1860): prep with camera_grad, render, masked L1 sums, backward",
                                   "fused_ms": time_tracking(True), "eager_around_same_rasterizer_ms": time_tracking(False),
                                   "fused_map_detached_ms": time_tracking(True, True),
                                   "note2": "map tensors detached -> geometry-only rasterizer backward (one 64-byte row per Gaussian, no semantic upstream gradients)"},
            "with_leaf_head": {"note": "mapping iterations >= 14 add the 1x1-conv leaf MLP + cross-entropy (scripts/hierslam.py:975-983)",
                               "fused_ms": timeit(True, True), "eager_around_same_rasterizer_ms": timeit(False, True),
                               "loss_fused": float(iteration(True, True).detach()), "loss_eager": float(iteration(False, True).detach())},
            "non_blocking_forward": run_ahead}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=500000)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    print(json.dumps(measure(a.P, iters=a.iters)))
