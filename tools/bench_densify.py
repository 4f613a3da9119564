#!/usr/bin/env python3
"""Times the device densification step (include/hsr_densify.h: non-presence mask incl. the median, order-preserving
compaction, back-projection) against the reference's torch expressions (scripts/hierslam.py:1271-1296, :144-194) on the same
device, 1200x680.  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hier-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(H=680, W=1200, iters=20):
    from hsr_utils import densify as D
    from test_densify import make_frame
    sil, rd, gt, col, K, w2c = make_frame(H, W, 2)
    t = lambda a: torch.tensor(a, device="cuda")
    sil, rd, gt, col, w2c = t(sil), t(rd), t(gt), t(col), t(w2c)
    Kt = torch.tensor(K)

    def fused():
        return D.non_presence_points(sil, rd, gt, col, Kt, w2c, 0.5)

    def eager():
        depth_error = torch.abs(gt - rd) * (gt > 0)
        mask = (sil < 0.5) | ((rd > gt) * (depth_error > 50 * depth_error.median()))
        mask = mask.reshape(-1)
        if torch.sum(mask) > 0:
            mask = mask & (gt > 0).reshape(-1)
            xg, yg = torch.meshgrid(torch.arange(W).cuda().float(), torch.arange(H).cuda().float(), indexing='xy')
            xx, yy = ((xg - K[0][2]) / K[0][0]).reshape(-1), ((yg - K[1][2]) / K[1][1]).reshape(-1)
            z = gt.reshape(-1)
            pts4 = torch.cat((torch.stack((xx * z, yy * z, z), dim=-1), torch.ones(H * W, 1).cuda().float()), dim=1)
            pts = (torch.inverse(w2c) @ pts4.T).T[:, :3]
            msd = (z / ((K[0][0] + K[1][1]) / 2)) ** 2
            cols = torch.permute(col, (1, 2, 0)).reshape(-1, 3)
            return torch.cat((pts, cols), -1)[mask], msd[mask]

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    a, b = fused(), eager()
    out = {"H": H, "W": W, "new_points": int(a[0].shape[0]), "same_count": int(a[0].shape[0]) == int(b[0].shape[0]),
           "max_abs_diff_xyz": float((a[0][:, :3] - b[0][:, :3]).abs().max()), "fused_ms": timeit(fused), "torch_eager_ms": timeit(eager)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
