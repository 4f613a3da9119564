#!/usr/bin/env python3
"""Times the fused loss heads (include/hsr_losses.h) — value + gradient w.r.t. the rendered maps — against the torch eager
chains of scripts/hierslam.py:921-974 on the same device, and the numpy oracle on the host.  One JSON line.

    python tools/bench_losses.py [--H 680 --W 1200] [--iters 30]

Workload: the mapping loss of a Replica frame — 0.8*L1 + 0.2*(1-SSIM) on the colour map, masked mean L1 on the depth map,
five-level cross-entropy on the K=26 logit planes — forward and backward to the three maps."""
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
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def measure(H=680, W=1200, iters=30, cpu=True):
    from hsr_utils import losses as L
    sizes = [2, 4, 6, 6, 8]
    K = sum(sizes)
    g = torch.Generator().manual_seed(0)
    im = torch.rand(3, H, W, generator=g).cuda().requires_grad_(True)
    depth = (torch.rand(1, H, W, generator=g) * 5).cuda().requires_grad_(True)
    sem = (torch.randn(K, H, W, generator=g) * 2).cuda().requires_grad_(True)
    gt_im, gt_d = torch.rand(3, H, W, generator=g).cuda(), (torch.rand(1, H, W, generator=g) * 5).cuda()
    gt_d[0, :20] = 0
    lab = torch.stack([torch.randint(0, n, (H, W), generator=g) for n in sizes]).cuda()
    C_leaf = 102
    mlp = torch.nn.Conv2d(K, C_leaf, kernel_size=1).cuda()
    leaf_lab = torch.randint(0, C_leaf, (H, W), generator=g).cuda()
    w1 = torch.tensor([np.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    w1 = (w1 / w1.sum()).unsqueeze(1)
    win = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0).expand(3, 1, 11, 11).contiguous().cuda()

    def eager_ssim(a, b):
        mu1, mu2 = F.conv2d(a, win, padding=5, groups=3), F.conv2d(b, win, padding=5, groups=3)
        s1 = F.conv2d(a * a, win, padding=5, groups=3) - mu1.pow(2)
        s2 = F.conv2d(b * b, win, padding=5, groups=3) - mu2.pow(2)
        s12 = F.conv2d(a * b, win, padding=5, groups=3) - mu1 * mu2
        return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1.pow(2) + mu2.pow(2) + 1e-4) * (s1 + s2 + 9e-4))).mean()

    def fused():
        mask = (gt_d > 0) & ~torch.isnan(depth)
        loss = L.mapping_image_loss(im, gt_im) + L.masked_l1(depth, gt_d, mask.detach(), "mean") + L.tree_cross_entropy(sem, lab, sizes)
        loss.backward()
        return loss

    def eager():
        mask = ((gt_d > 0) & ~torch.isnan(depth)).detach()
        ce, b = 0.0, 0
        celoss = torch.nn.CrossEntropyLoss()
        for i, n in enumerate(sizes):
            lvl = sem[b:b + n].permute(1, 2, 0)
            ce = ce + celoss(lvl.reshape(-1, n), lab[i].view(-1).long())
            b += n
        loss = 0.8 * torch.abs(im - gt_im).mean() + 0.2 * (1.0 - eager_ssim(im, gt_im)) + torch.abs(gt_d - depth)[mask].mean() + ce
        loss.backward()
        return loss

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            for v in (im, depth, sem, *mlp.parameters()):
                v.grad = None
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    lf, le = float(fused().detach()), float(eager().detach())
    N = H * W
    out = {"H": H, "W": W, "K": K, "levels": sizes, "fused_ms": timeit(fused), "torch_eager_ms": timeit(eager),
           "loss_fused": lf, "loss_eager": le,
           # maps read + gradients written: colour L1 3*3, SSIM fwd 3*(2+3) + bwd 3*(3+2+1), depth 3+1, CE K*3 + 8*levels
           "alg_bytes": N * (4 * (9 + 15 + 18 + 3 + 3 * K) + 1 + 8 * len(sizes))}
    out["fused_alg_GBps"] = out["alg_bytes"] / (out["fused_ms"] * 1e-3) / 1e9
    # leaf head (scripts/hierslam.py:976-983): 1x1-conv MLP K -> 102 classes + cross-entropy; the conv stays torch (rocBLAS /
    # MIOpen GEMM), the CE on its planar output is the fused kernel (no [H*W, C] permute copy)
    def leaf_fused():
        loss = L.cross_entropy_planar(mlp(sem.unsqueeze(0)), leaf_lab)
        loss.backward()
        return loss

    def leaf_eager():
        logits = mlp(sem.unsqueeze(0))
        logits = logits.squeeze(0).view(logits.shape[1], -1).permute(1, 0)
        loss = torch.nn.CrossEntropyLoss()(logits, leaf_lab.view(-1).long())
        loss.backward()
        return loss
    def leaf_fully_fused():
        loss = L.leaf_mlp_cross_entropy(sem, mlp, leaf_lab)
        loss.backward()
        return loss
    out["leaf_head_fused_ms"] = timeit(leaf_fully_fused)
    out["leaf_loss_fully_fused"] = float(leaf_fully_fused().detach())
    out["leaf_head_fused_ce_ms"] = timeit(leaf_fused)
    out["leaf_head_eager_ms"] = timeit(leaf_eager)
    out["leaf_loss_fused"], out["leaf_loss_eager"] = float(leaf_fused().detach()), float(leaf_eager().detach())
    if cpu:
        import loss_oracle as LO
        t0 = time.perf_counter()
        LO.ssim(im.detach().cpu().numpy(), gt_im.cpu().numpy())
        LO.l1_mean(im.detach().cpu().numpy(), gt_im.cpu().numpy())
        LO.masked_l1(depth.detach().cpu().numpy(), gt_d.cpu().numpy(), (gt_d > 0).cpu().numpy(), "mean")
        LO.tree_cross_entropy(sem.detach().cpu().numpy(), lab.cpu().numpy(), sizes)
        out["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=680)
    ap.add_argument("--W", type=int, default=1200)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    print(json.dumps(measure(a.H, a.W, a.iters, not a.no_cpu)))
