#!/usr/bin/env python3
"""Not a test: the CPU simulation behind DESIGN.md §4 "4x4 sub-block lists".  Takes the geometry state of the headline scene
from the oracle (test infrastructure, hence this file lives under tests/) and counts, per tile-kernel granularity, how many
(splat, cell) visits the alpha >= 1/255 bounding boxes generate and how many iterations a wave would run:
  * quadrant lists (8x8 cells, all 64 lanes visit every entry),
  * flat 4x4 sub-block lists (forward: iterations = max over the wave's four groups),
  * 4x4 sub-block masks over 16-entry chunks of the quadrant list (backward).
Usage: python tests/sim_sublists.py [P]   (default 500000; ~1 minute, ~4 GB)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "hier-slam_amd")]
import oracle_lib as O  # noqa: E402
from hsr_utils.camera import replica_intrinsics, setup_camera_tensors  # noqa: E402
from hsr_utils.synthetic import make_scene  # noqa: E402

W, H, K = 1200, 680, 26
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
k = replica_intrinsics(W, H)
cam = setup_camera_tensors(W, H, k, np.eye(4))
sc = make_scene(P, W, H, K, k, seed=0)
_, st = O.forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors_precomp"], semantics_precomp=sc["semantics_precomp"],
                  scales=sc["scales"], rotations=sc["rotations"])
m2, co, rad = st.field("means2D"), st.field("conic_opacity"), st.field("radii")
vis = np.nonzero(rad > 0)[0]
x, y, r = m2[vis, 0], m2[vis, 1], rad[vis].astype(np.float32)
A, B, C, o = co[vis, 0], co[vis, 1], co[vis, 2], co[vis, 3]
t = np.log(np.maximum(255 * o, 1e-9))
ok = t > 0
det = np.maximum(A * C - B * B, 1e-12)
hx = np.minimum(np.sqrt(np.maximum(2 * t * C / det, 0)), r)   # half extents of the alpha >= 1/255 ellipse (quadrant_mask)
hy = np.minimum(np.sqrt(np.maximum(2 * t * A / det, 0)), r)
x, y, hx, hy = x[ok], y[ok], hx[ok], hy[ok]
print("visible %d, num_rendered %d, splats with 255*opacity > 1: %d" % (len(vis), st.R, ok.sum()))

g, nx, ny = 4, (W + 3) // 4, (H + 3) // 4
ax = np.clip(np.floor((x - hx) / g), 0, nx).astype(np.int64); bx = np.clip(np.floor((x + hx) / g) + 1, 0, nx).astype(np.int64)
ay = np.clip(np.floor((y - hy) / g), 0, ny).astype(np.int64); by = np.clip(np.floor((y + hy) / g) + 1, 0, ny).astype(np.int64)
w, h = bx - ax, by - ay
n = w * h
idx = np.repeat(np.arange(len(x)), n)
off = np.arange(n.sum()) - np.repeat(np.cumsum(n) - n, n)
cx, cy = ax[idx] + off % np.maximum(w[idx], 1), ay[idx] + off // np.maximum(w[idx], 1)
nqx, nqy = (nx + 1) // 2, (ny + 1) // 2
qid, bit = (cy // 2) * nqx + (cx // 2), (cy % 2) * 2 + (cx % 2)
key = idx.astype(np.int64) * (nqx * nqy) + qid
order = np.argsort(key, kind="stable")
key, bit = key[order], bit[order]
uk = np.unique(key)
mask = np.zeros(len(uk), np.int64)
np.bitwise_or.at(mask, np.searchsorted(uk, key), 1 << bit)
quad = uk % (nqx * nqy)
print("quadrant-list entries %d (x64 lanes = %d lane visits); sub-block entries %d (x16 = %d)" % (len(uk), 64 * len(uk), len(key), 16 * len(key)))
rng = np.random.default_rng(0)   # depth order within a quadrant is independent of geometry in this scene
perm = rng.permutation(len(uk))
quad, mask = quad[perm], mask[perm]
order = np.argsort(quad, kind="stable")
quad, mask = quad[order], mask[order]
first = np.r_[0, np.nonzero(np.diff(quad))[0] + 1]
cnt = np.diff(np.r_[first, len(quad)])
pos = np.arange(len(quad)) - np.repeat(first, cnt)
for chunk in (16, 32, 1 << 30):
    ck = quad * 100000 + np.minimum(pos // chunk, 99999)
    u, inv = np.unique(ck, return_inverse=True)
    c = np.zeros((len(u), 4), np.int64)
    for b in range(4):
        np.add.at(c[:, b], inv, (mask >> b) & 1)
    it = c.max(axis=1).sum()
    print("chunk %-10s wave iterations %d = %.2f x the quadrant list" % ("unbounded" if chunk > 1 << 20 else chunk, it, it / len(quad)))
