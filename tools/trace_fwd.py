#!/usr/bin/env python3
"""Phase timers of the default forward tile kernel (diagnostic build: make -C hier-slam_amd/csrc trace).
Per wave (= tile quadrant): shader cycles in the prologue (ranges -> ids -> first records -> first barrier), inside workgroup
barriers, in the staging work between them, in the blend loops, in the epilogue (output stores).
Usage:  python tools/trace_fwd.py   (loads hier-slam_amd/libhsr_rast_trace.so)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HSR_RAST_LIB", os.path.join(ROOT, "hier-slam_amd", "libhsr_rast_trace.so"))
os.environ["HSR_GLUE"] = "ctypes"
sys.path.insert(0, os.path.join(ROOT, "hier-slam_amd"))
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer_semantic, _C  # noqa: E402
from hsr_utils.camera import replica_intrinsics, setup_camera_tensors  # noqa: E402
from hsr_utils.synthetic import make_scene  # noqa: E402

W, H, K, P = 1200, 680, int(os.environ.get("K", 26)), 500000
k = replica_intrinsics(W, H)
cam_cpu = setup_camera_tensors(W, H, k, np.eye(4))
dev = torch.device("cuda")
cam = GaussianRasterizationSettings(**{kk: (v.to(dev) if isinstance(v, torch.Tensor) else v) for kk, v in cam_cpu.items()})
sc = make_scene(P, W, H, K, k, seed=0, kind=os.environ.get("KIND", "slam"))
leaf = {n: sc[n].to(dev) for n in ("means3D", "colors_precomp", "semantics_precomp", "opacities", "scales", "rotations")}
r = GaussianRasterizer_semantic(cam)
with torch.no_grad():
    for _ in range(3):
        r(means3D=leaf["means3D"], means2D=torch.zeros(P, 3, device=dev), opacities=leaf["opacities"], colors_precomp=leaf["colors_precomp"],
          scales=leaf["scales"], rotations=leaf["rotations"], semantics_precomp=leaf["semantics_precomp"])
torch.cuda.synchronize()
T = ((W + 15) // 16) * ((H + 15) // 16)
n = min(T * 4, 16384) * 8
buf = (C.c_ulonglong * n)()
rc = _C._lib.hsr_debug_read_trace_fwd(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
a = a[a[:, 0] > 0]
tot = a[:, 0].mean()
print("render_fwd_kernel (sub-block lists), K = %d: waves traced %d, rc %d" % (K, a.shape[0], rc))
for i, nm in enumerate(["total", "prologue", "in barriers", "staging work", "blend loops", "epilogue", "iterations", "list publish"]):
    print("%-14s mean %10.0f   p10 %10.0f   p90 %10.0f   max %10.0f   (%.1f %% of wave life)" % (
        nm, a[:, i].mean(), np.percentile(a[:, i], 10), np.percentile(a[:, i], 90), a[:, i].max(), 100 * a[:, i].mean() / tot))
print("cycles per iteration in the blend loops: %.0f" % (a[:, 4].sum() / a[:, 6].sum()))
tiles = a.shape[0] // 4
q = a[: tiles * 4, 4].reshape(tiles, 4)
print("blend cycles: mean over waves %.0f, mean of per-tile max %.0f (imbalance x%.2f)" % (q.mean(), q.max(axis=1).mean(), q.max(axis=1).mean() / q.mean()))
print("(list publish is part of the staging work; the rest of it is the sub-block mask, the record pre-scaling and the LDS stores)")
