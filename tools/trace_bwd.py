#!/usr/bin/env python3
"""Phase timers of the default backward tile kernel (diagnostic build: make -C hier-slam_amd/csrc trace).
Runs the headline workload through libhsr_rast_trace.so and prints, per wave (= tile quadrant), the average shader cycles spent
in: prologue, staging + barriers, the blend loop (excluding flushes), matrix-core flushes, row emission (atomics), and the
visit counts.  Usage:  HSR_RAST_LIB=hier-slam_amd/libhsr_rast_trace.so python tools/trace_bwd.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HSR_RAST_LIB", os.path.join(ROOT, "hier-slam_amd", "libhsr_rast_trace.so"))
os.environ["HSR_GLUE"] = "ctypes"   # the compiled glue is linked against the product library, not the trace build
sys.path.insert(0, os.path.join(ROOT, "hier-slam_amd"))
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer_semantic, _C  # noqa: E402
from hsr_utils.camera import replica_intrinsics, setup_camera_tensors  # noqa: E402
from hsr_utils.synthetic import make_scene, make_upstream_grads  # noqa: E402

W, H, K, P = 1200, 680, int(os.environ.get("TRACE_K", "26")), 500000
GEO = os.environ.get("TRACE_GEO", "0") == "1"   # tracking iteration: gradients for the means only
k = replica_intrinsics(W, H)
cam_cpu = setup_camera_tensors(W, H, k, np.eye(4))
dev = torch.device("cuda")
cam = GaussianRasterizationSettings(**{kk: (v.to(dev) if isinstance(v, torch.Tensor) else v) for kk, v in cam_cpu.items()})
sc = make_scene(P, W, H, K, k, seed=0)
up = make_upstream_grads(W, H, K, seed=1)
upd = [up[n].to(dev) for n in ("color", "semantic", "depth", "median", "opacity")]
names = ("means3D", "colors_precomp", "semantics_precomp", "opacities", "scales", "rotations")
leaf = {n: sc[n].to(dev).requires_grad_((not GEO) or n == "means3D") for n in names}
r = GaussianRasterizer_semantic(cam)
for _ in range(3):
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    outs = r(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], colors_precomp=leaf["colors_precomp"], scales=leaf["scales"],
             rotations=leaf["rotations"], semantics_precomp=leaf["semantics_precomp"])
    torch.autograd.backward([outs[0], outs[2], outs[3], outs[4], outs[5]], upd)
torch.cuda.synchronize()
T = ((W + 15) // 16) * ((H + 15) // 16)
n = min(T * 4, 16384) * 8
buf = (C.c_ulonglong * n)()
impl = os.environ.get("HSR_BWD_IMPL", "")
sub = impl != "mfma"   # default backward: round 4's Q-panel kernel; HSR_BWD_IMPL=sub: round 3's butterfly kernel; mfma: the quadrant-list one
qk = impl == ""
rc = (_C._lib.hsr_debug_read_trace_q if qk else (_C._lib.hsr_debug_read_trace_sub if sub else _C._lib.hsr_debug_read_trace))(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
a = a[a[:, 0] > 0]
tot = a[:, 0].mean()
print("kernel:", ("render_bwd_q_kernel" if qk else "render_bwd_sub_kernel") if sub else "render_bwd_mfma_kernel", "K", K, "geo", GEO, " waves traced:", a.shape[0], "rc", rc)
if sub:
    names = ["total", "prologue", "stage+barriers", "loop(incl. flush)", "flush", "chunks", "wave iterations" if qk else "group visits", "chunk set-up (in loop)" if qk else "... with a pixel that accepts"]
    for i, nm in enumerate(names):
        print("%-18s mean %10.0f   p10 %10.0f   p90 %10.0f   max %10.0f" % (nm, a[:, i].mean(), np.percentile(a[:, i], 10), np.percentile(a[:, i], 90), a[:, i].max()))
    print("fractions of wave time: prologue %.2f  stage %.2f  loop %.2f (of which flush + emission %.2f)" % (
        a[:, 1].mean() / tot, a[:, 2].mean() / tot, a[:, 3].mean() / tot, a[:, 4].mean() / tot))
    if qk:
        print("per chunk: set-up %.0f, visits %.0f, flush + emission %.0f cycles ; stage-A evaluations per chunk: %.2f ; visit-loop cycles per stage-A evaluation: %.0f" % (
            a[:, 7].sum() / a[:, 5].sum(), (a[:, 3] - a[:, 4] - a[:, 7]).sum() / a[:, 5].sum(), a[:, 4].sum() / a[:, 5].sum(), a[:, 6].sum() / a[:, 5].sum(),
            (a[:, 3] - a[:, 4] - a[:, 7]).sum() / a[:, 6].sum()))
    else:
        print("flush + emission cycles per chunk: %.0f ; (16-lane group, splat) visits per chunk: %.1f ; fraction of them in which a pixel accepts the splat: %.3f" % (
            a[:, 4].sum() / a[:, 5].sum(), a[:, 6].sum() / a[:, 5].sum(), a[:, 7].sum() / a[:, 6].sum()))
else:
    names = ["total", "prologue", "stage+barriers", "loop(no flush)", "flush", "emit", "visits", "accepted"]
    for i, nm in enumerate(names):
        print("%-16s mean %10.0f   p10 %10.0f   p90 %10.0f   max %10.0f" % (nm, a[:, i].mean(), np.percentile(a[:, i], 10), np.percentile(a[:, i], 90), a[:, i].max()))
    print("fractions of wave time: prologue %.2f  stage %.2f  loop %.2f (of which flush %.2f, emit %.2f)" % (
        a[:, 1].mean() / tot, a[:, 2].mean() / tot, a[:, 3].mean() / tot, a[:, 4].mean() / tot, a[:, 5].mean() / tot))
    print("cycles per visit in loop (excl. flush): %.0f ; per accepted: %.0f ; flush cycles per 16 accepted: %.0f" % (
        (a[:, 3] - a[:, 4]).sum() / a[:, 6].sum(), (a[:, 3] - a[:, 4]).sum() / a[:, 7].sum(), 16 * a[:, 4].sum() / a[:, 7].sum()))
# per-tile imbalance: max over the four quadrant waves vs their mean
tiles = a.shape[0] // 4
q = a[: tiles * 4, 3].reshape(tiles, 4)
print("loop cycles: mean over waves %.0f, mean of per-tile max %.0f (imbalance x%.2f)" % (q.mean(), q.max(axis=1).mean(), q.max(axis=1).mean() / q.mean()))
