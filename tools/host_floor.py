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
