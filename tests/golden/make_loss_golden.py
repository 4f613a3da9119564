"""Generates tests/golden/loss_*.npz: inputs, outputs and autograd gradients of the REFERENCE's own device-free loss helpers
(l1_loss_v1: utils/slam_helpers.py:5-6; calc_ssim: utils/slam_external.py:66-97), imported from /root/reference in the
build container, plus torch.nn.CrossEntropyLoss over tree levels exactly as scripts/hierslam.py:963-974 applies it.
Only data is stored (no reference source).  Run: python tests/golden/make_loss_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from utils.slam_external import calc_ssim  # noqa: E402
from utils.slam_helpers import l1_loss_v1  # noqa: E402


def smooth_image(g, C, H, W):
    """natural-image-like: low-frequency content + mild noise, in [0, 1]"""
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.zeros((C, H, W))
    for c in range(C):
        for _ in range(6):
            fx, fy, ph = g.uniform(0.01, 0.15), g.uniform(0.01, 0.15), g.uniform(0, 6.28)
            img[c] += g.uniform(0.05, 0.25) * np.sin(fx * xx + fy * yy + ph)
    img = 0.5 + img + g.normal(0, 0.02, img.shape)
    return np.clip(img, 0, 1).astype(np.float32)


def main():
    g = np.random.default_rng(7)
    cases = {"random_40x56": (g.random((3, 40, 56)).astype(np.float32), g.random((3, 40, 56)).astype(np.float32))}
    a = smooth_image(g, 3, 68, 120)
    cases["smooth_68x120"] = (np.clip(a + g.normal(0, 0.03, a.shape), 0, 1).astype(np.float32), a)
    cases["tiny_7x9"] = (g.random((3, 7, 9)).astype(np.float32), g.random((3, 7, 9)).astype(np.float32))  # smaller than the window
    out = {}
    for name, (x, y) in cases.items():
        tx = torch.tensor(x, requires_grad=True)
        ty = torch.tensor(y)
        s = calc_ssim(tx, ty)
        s.backward()
        out[name + "/img1"], out[name + "/img2"] = x, y
        out[name + "/ssim"], out[name + "/ssim_grad"] = s.detach().numpy(), tx.grad.numpy().copy()
        tx.grad = None
        l = l1_loss_v1(tx, ty)
        l.backward()
        out[name + "/l1"], out[name + "/l1_grad"] = l.detach().numpy(), tx.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "loss_ssim_l1.npz"), **out)

    # tree cross-entropy: Replica-like 5-level tree, K = 26 (scripts/hierslam.py:963-974 applied level by level)
    level_sizes = [2, 4, 6, 6, 8]
    K, H, W = sum(level_sizes), 24, 40
    logits = (g.normal(0, 2.0, (K, H, W))).astype(np.float32)
    labels = np.stack([g.integers(0, n, (H, W)) for n in level_sizes] + [g.integers(0, 30, (H, W))]).astype(np.int64)
    labels[1, :3, :5] = -100   # CrossEntropyLoss's default ignore_index
    tz = torch.tensor(logits, requires_grad=True)
    ce = torch.nn.CrossEntropyLoss()
    total, per_level, begin = 0.0, [], 0
    for i, n in enumerate(level_sizes):
        lvl = tz[begin:begin + n].permute(1, 2, 0)
        lvl = lvl.view(-1, lvl.size(2))
        li = ce(lvl, torch.tensor(labels[i]).view(-1).long())
        per_level.append(float(li))
        total = total + li
        begin += n
    total.backward()
    np.savez_compressed(os.path.join(HERE, "loss_tree_ce.npz"), logits=logits, labels=labels, level_sizes=np.array(level_sizes),
                        per_level=np.array(per_level), grad=tz.grad.numpy())
    # leaf head: torch.nn.Conv2d(K, C, kernel_size=1) + CrossEntropyLoss exactly as scripts/hierslam.py:976-983 chains them
    K, C, H, W = 26, 102, 20, 36
    sem = g.normal(0, 1.5, (K, H, W)).astype(np.float32)
    mlp = torch.nn.Conv2d(K, C, kernel_size=1)
    with torch.no_grad():
        mlp.weight.copy_(torch.tensor(g.normal(0, 0.3, (C, K, 1, 1)).astype(np.float32)))
        mlp.bias.copy_(torch.tensor(g.normal(0, 0.2, (C,)).astype(np.float32)))
    lab = g.integers(0, C, (H, W)).astype(np.int64)
    lab[:2, :7] = -100
    ts = torch.tensor(sem, requires_grad=True)
    logits = mlp(ts.unsqueeze(0))
    logits = logits.squeeze(0).view(logits.shape[1], -1).permute(1, 0)
    loss = torch.nn.CrossEntropyLoss()(logits, torch.tensor(lab).view(-1).long())
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "loss_leaf_mlp.npz"), sem=sem, weight=mlp.weight.detach().numpy().reshape(C, K),
                        bias=mlp.bias.detach().numpy(), labels=lab, loss=loss.detach().numpy(), d_sem=ts.grad.numpy(),
                        d_weight=mlp.weight.grad.numpy().reshape(C, K), d_bias=mlp.bias.grad.numpy())
    print("written", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
