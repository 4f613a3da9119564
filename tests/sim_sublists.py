#!/usr/bin/env python3
"""Not a test: the CPU count behind DESIGN.md §4b "exact ellipse-vs-sub-block culling".  Takes the geometry state and the sorted
instance list of a headline-sized scene from the oracle (test infrastructure, hence this file lives under tests/), re-evaluates
hsr_tile_common.h's subblock_mask in numpy for every (tile, Gaussian) instance — once with the bounding-box test of round 1, once
with the exact row-slab test of round 2 — and prints how many sub-block entries each keeps and how many instances end up with an
empty mask.  Usage: python tests/sim_sublists.py [P] [slam|aniso]   (default 500000 slam; ~1 minute, ~4 GB)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "hier-slam_amd")]
import oracle_lib as O  # noqa: E402
from hsr_utils.camera import replica_intrinsics, setup_camera_tensors  # noqa: E402
from hsr_utils.synthetic import make_scene  # noqa: E402

W, H, K = 1200, 680, 4
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
kind = sys.argv[2] if len(sys.argv) > 2 else "slam"
k = replica_intrinsics(W, H)
cam = setup_camera_tensors(W, H, k, np.eye(4))
sc = make_scene(P, W, H, K, k, seed=0, kind=kind)
_, st = O.forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors_precomp"], semantics_precomp=sc["semantics_precomp"],
                  scales=sc["scales"], rotations=sc["rotations"])
m2, co = st.field("means2D"), st.field("conic_opacity")
keys, vals = st.field("keys"), st.field("vals")
tiles = (keys >> np.uint64(32)).astype(np.int64)
g = vals.astype(np.int64)
tiles_x = (W + 15) // 16
f = np.float32
tx, ty = ((tiles % tiles_x) * 16).astype(f), ((tiles // tiles_x) * 16).astype(f)
x, y = m2[g, 0].astype(f), m2[g, 1].astype(f)
A, B, C, o = [co[g, i].astype(f) for i in range(4)]
t255 = f(255) * o
ok = t255 >= 1
tau = f(2) * np.log(np.maximum(t255, 1)).astype(f) * f(1.002) + f(0.02)
det = A * C - B * B
inv_det, inv_a = f(1) / det, f(1) / A
hx = np.sqrt(tau * C * inv_det)
hy = np.sqrt(tau * A * inv_det) * f(1.001) + f(0.02)
dyp, atau, nb = -(B / C) * hx, A * tau, -B
rx, ry = x - tx, y - ty
hxb = hx * f(1.001) + f(0.05)
n_exact = np.zeros(len(g), np.int64)
n_box = np.zeros(len(g), np.int64)
for r in range(4):
    lo, hi = np.maximum(ry - f(4 * r + 3), -hy), np.minimum(ry - f(4 * r), hy)
    top = np.maximum(lo, hi)
    dyu, dyl = np.clip(dyp, lo, top), np.clip(-dyp, lo, top)
    xmax = (nb * dyu + np.sqrt(np.maximum(atau - det * dyu * dyu, 0))) * inv_a + f(0.02)
    xmin = (nb * dyl - np.sqrt(np.maximum(atau - det * dyl * dyl, 0))) * inv_a - f(0.02)
    row_on = (lo <= hi) & ok
    for c in range(4):
        n_exact += row_on & ((rx - f(4 * c + 3)) <= xmax) & ((rx - f(4 * c)) >= xmin)
        n_box += row_on & ((rx - f(4 * c + 3)) <= hxb) & ((rx - f(4 * c)) >= -hxb)
print("%s scene, P = %d: %d (tile, Gaussian) instances" % (kind, P, len(g)))
print("sub-block entries: bounding box %d, exact %d (x%.3f); instances with an empty mask: box %d, exact %d" % (
    n_box.sum(), n_exact.sum(), n_exact.sum() / n_box.sum(), (n_box == 0).sum(), (n_exact == 0).sum()))
