#!/usr/bin/env python3
"""Times the fused rasterizer-input preparation (include/hsr_frame_prep.h) forward+backward against the torch eager
op chain it replaces (same device), and the numpy oracle on the host.  One JSON line.

    python tools/bench_frame_prep.py [--P 500000] [--S 1] [--iters 50]

Algorithmic bytes per Gaussian (S=1, semantic variant): forward reads 12+16+4+4 and writes 12+16+16+4+12 = 96 B;
backward reads the same 36 B of inputs + 12+16+4+12 of upstream gradients and writes 12+16+4+4 = 116 B."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hier-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def measure(P=500000, S=1, iters=50, cpu=True):
    import torch.nn.functional as F
    from hsr_utils import slam_helpers as SH
    from test_frame_prep import make_inputs
    inp = make_inputs(P, S, frames=8)
    t = {k: torch.tensor(v, device="cuda", requires_grad=True) for k, v in inp.items()}
    t["rgb_colors"] = torch.rand(P, 3, device="cuda")
    t["semantic"] = torch.rand(P, 26, device="cuda")
    up = {k: torch.randn(P, c, device="cuda") for k, c in (("means3D", 3), ("rotations", 4), ("opacities", 1), ("scales", 3))}

    def fused():
        rv = SH.transformed_params2rendervar_semantic(t, SH.transform_to_frame(t, 3, True, True))
        torch.autograd.backward([rv[k] for k in up], [up[k] for k in up])

    def eager():
        q = F.normalize(t['cam_unnorm_rots'][..., 3])
        n = q / torch.sqrt((q * q).sum(dim=1))[:, None]
        r, x, y, z = n[:, 0], n[:, 1], n[:, 2], n[:, 3]
        rot = torch.zeros((1, 3, 3), device="cuda")
        rot[:, 0, 0] = 1 - 2 * (y * y + z * z); rot[:, 0, 1] = 2 * (x * y - r * z); rot[:, 0, 2] = 2 * (x * z + r * y)
        rot[:, 1, 0] = 2 * (x * y + r * z); rot[:, 1, 1] = 1 - 2 * (x * x + z * z); rot[:, 1, 2] = 2 * (y * z - r * x)
        rot[:, 2, 0] = 2 * (x * z - r * y); rot[:, 2, 1] = 2 * (y * z + r * x); rot[:, 2, 2] = 1 - 2 * (x * x + y * y)
        rel = torch.eye(4, device="cuda"); rel[:3, :3] = rot[0]; rel[:3, 3] = t['cam_trans'][0, :, 3]
        pts4 = torch.cat((t['means3D'], torch.ones(P, 1, device="cuda")), dim=1)
        rv = {'means3D': (rel @ pts4.T).T[:, :3], 'rotations': F.normalize(t['unnorm_rotations']),
              'opacities': torch.sigmoid(t['logit_opacities']), 'scales': torch.exp(torch.tile(t['log_scales'], (1, 3)) if S == 1 else t['log_scales'])}
        torch.autograd.backward([rv[k] for k in up], [up[k] for k in up])

    def timeit(fn):
        for _ in range(5):
            fn()
        for v in t.values():
            v.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
            for v in t.values():
                v.grad = None
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    out = {"P": P, "S": S, "fused_ms": timeit(fused), "torch_eager_ms": timeit(eager)}
    out["alg_bytes"] = P * ((36 + 2 * (S - 1) * 4) + 60 + 36 + 44 + 36)
    out["fused_alg_GBps"] = out["alg_bytes"] / (out["fused_ms"] * 1e-3) / 1e9
    if cpu:
        import frame_prep_oracle as O
        g = {k: v.cpu().numpy() for k, v in up.items()}
        t0 = time.perf_counter()
        O.forward(**inp, time_idx=3)
        O.backward(**inp, time_idx=3, grads=g)
        out["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=500000)
    ap.add_argument("--S", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    print(json.dumps(measure(a.P, a.S, a.iters)))
