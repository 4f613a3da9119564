#!/usr/bin/env python3
"""Host-side floor of one fwd+bwd render: the same autograd call sequence as bench.py on a tiny scene (GPU work ~ nothing), so the
step time is what the host needs to enqueue a render (Python glue + C ABI + the num_rendered read-back round trip)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hier-slam_amd"))
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer_semantic  # noqa: E402
from hsr_utils.camera import replica_intrinsics, setup_camera_tensors  # noqa: E402
from hsr_utils.synthetic import make_scene, make_upstream_grads  # noqa: E402
W, H, K, P = 64, 48, 26, 2000
k = replica_intrinsics(W, H)
cam_cpu = setup_camera_tensors(W, H, k, np.eye(4))
dev = torch.device("cuda")
cam = GaussianRasterizationSettings(**{kk: (v.to(dev) if isinstance(v, torch.Tensor) else v) for kk, v in cam_cpu.items()})
sc = make_scene(P, W, H, K, k, seed=0)
up = make_upstream_grads(W, H, K, seed=1)
upd = [up[n].to(dev) for n in ("color", "semantic", "depth", "median", "opacity")]
names = ("means3D", "colors_precomp", "semantics_precomp", "opacities", "scales", "rotations")
leaf = {n: sc[n].to(dev).requires_grad_(True) for n in names}
r = GaussianRasterizer_semantic(cam)
def step():
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    outs = r(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], colors_precomp=leaf["colors_precomp"], scales=leaf["scales"],
             rotations=leaf["rotations"], semantics_precomp=leaf["semantics_precomp"])
    torch.autograd.backward([outs[0], outs[2], outs[3], outs[4], outs[5]], upd)
for _ in range(20): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize()
    print("host floor: %.3f ms per fwd+bwd render" % ((time.perf_counter() - t0) / 200 * 1e3))
if os.environ.get("HSR_HOST_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300): step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
if os.environ.get("HSR_HOST_CALLS"):
    # time spent inside the two C ABI calls (ctypes glue: run with HSR_GLUE=ctypes)
    from diff_gaussian_rasterization import _C
    acc = {"fwd": 0.0, "bwd": 0.0}
    f0, b0 = _C._lib.hsr_forward_semantic, _C._lib.hsr_backward_semantic
    def fw(*a):
        t = time.perf_counter(); r = f0(*a); acc["fwd"] += time.perf_counter() - t; return r
    def bw(*a):
        t = time.perf_counter(); r = b0(*a); acc["bwd"] += time.perf_counter() - t; return r
    _C._lib.hsr_forward_semantic, _C._lib.hsr_backward_semantic = fw, bw
    t0 = time.perf_counter()
    for _ in range(300): step()
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 300 * 1e3
    print("per step %.3f ms: inside hsr_forward_semantic %.3f ms, inside hsr_backward_semantic %.3f ms" % (tot, acc["fwd"] / 300 * 1e3, acc["bwd"] / 300 * 1e3))
