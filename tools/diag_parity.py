"""Diagnostic: where do the largest HIP-vs-oracle gradient differences sit at the headline workload?"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "hier-slam_amd"), os.path.join(ROOT, "tests")]
import harness
from hsr_utils.camera import replica_intrinsics, setup_camera_tensors
from hsr_utils.synthetic import make_scene, make_upstream_grads
W, H, P, K = 1200, 680, int(os.environ.get("P", 500000)), 26
kmat = replica_intrinsics(W, H)
cam = setup_camera_tensors(W, H, kmat, np.eye(4))
sc = make_scene(P, W, H, K, kmat, seed=0, kind=os.environ.get("KIND", "slam"))
up = {n: v * float(W * H) for n, v in make_upstream_grads(W, H, K, seed=1).items()}
og, gg, sg = harness.run_gpu(cam, sc, up)
og2, gg2, _ = harness.run_gpu(cam, sc, up)
oo, go, so = harness.run_oracle(cam, sc, up)
for n in go:
    e = np.abs(gg[n].astype(np.float64) - go[n]); mx = np.abs(go[n]).max()
    e2 = np.abs(gg[n].astype(np.float64) - gg2[n])
    print("%-18s max|exp| %.3e  err/max %.2e   run-to-run (HIP vs HIP) %.2e   #elements > 1e-4*max: %d, > 3e-5*max: %d" % (
        n, mx, e.max() / max(mx, 1e-30), e2.max() / max(mx, 1e-30), int((e > 1e-4 * mx).sum()), int((e > 3e-5 * mx).sum())))
e = np.abs(gg["means3D"].astype(np.float64) - go["means3D"])
mx = np.abs(go["means3D"]).max()
worst = np.argsort(-e.max(1))[:12]
cov = so.field("cov3D"); co = so.field("conic_opacity"); m2 = so.field("means2D"); dep = so.field("depths")
for i in worst:
    det_inv = co[i, 0] * co[i, 2] - co[i, 1] ** 2
    print("g %7d  err %s  exp %s  got %s\n          z %.4f radius %d scale %s opacity %.3f conic %s (1/det %.3e)  m2 %s\n          grad means2D exp %s got %s | scales exp %s got %s" % (
        i, e[i], go["means3D"][i], gg["means3D"][i], dep[i], oo["radii"][i], sc["scales"][i].numpy(), float(sc["opacities"][i]), co[i, :3], det_inv, m2[i],
        go["means2D"][i, :2], gg["means2D"][i, :2], go["scales"][i], gg["scales"][i]))
