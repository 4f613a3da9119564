#!/usr/bin/env python3
"""How far can FMA contraction move the rasterizer's integer outputs?  (SURVEY.md §7 hard part 1, §8c.)

The reference is built by plain nvcc (RAST/setup.py:21-29: no -fmad=false), which fuses a*b+c pairs of its choosing; everything that
decides an integer — p_view / p_hom (auxiliary.h:41-66), cov2D -> radius (forward.cu:74-113, :199, :219-236) — is therefore evaluated
with SOME contraction pattern that cannot be known here.  This tool runs the CPU oracle twice on the same scene — the IEEE-order build
every parity test uses (-ffp-contract=off) and the same source with the compiler free to contract (-mfma -ffp-contract=fast,
oracle/libhsr_oracle_fma.so) — and counts what differs: radii, tiles_touched, depth key bits, num_rendered, tile ranges, per-tile
list orders, n_contrib, and the distance between the two builds' images and gradients.  CPU only.

usage: python tools/fma_sensitivity.py [--scenes goldens small headline stress] [--json out.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "hier-slam_amd"), os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402
import scenes  # noqa: E402

GOLDEN = {  # tests/golden/make_golden.py
    "golden sem_k26_aniso_64x48": (64, 48, 300, 26, "aniso", 2.5, True, (0.0, 0.0, 0.0), 21, 0.1),
    "golden sem_k16_slam_80x56_bg": (80, 56, 400, 16, "slam", 3.0, True, (0.2, 0.4, 0.6), 21, 0.1),
    "golden plain_mask_72x40": (72, 40, 300, 0, "aniso", 2.5, False, (0.0, 0.0, 0.0), 21, 0.1),
}
SMALL = {"small 256x256 P=5000 K=26 (BASELINE configs[0] size)": (256, 256, 5000, 26, "slam", 2.0, True, (0.0, 0.0, 0.0), 3, 0.0)}


def bench_scene(W, H, P, K, kind, rank=0):
    """the scene bench.py times (same generator, seeds and camera); rank > 0: the camera of that rank's keyframe in the multi-GPU bench
    (a small rotation + translation: with the identity camera of rank 0 the view transform multiplies by 0 and 1 only and the depth
    key bits cannot move)"""
    from hsr_utils.camera import replica_intrinsics, setup_camera_tensors
    from hsr_utils.synthetic import make_scene, make_upstream_grads
    k = replica_intrinsics(W, H)
    w2c = np.eye(4)
    if rank:
        a = 0.01 * rank
        w2c[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        w2c[:3, 3] = [0.01 * rank, -0.005 * rank, 0.0]
    cam = setup_camera_tensors(W, H, k, w2c)
    sc = make_scene(P, W, H, K, k, seed=0, kind=kind)
    up = make_upstream_grads(W, H, K, seed=1)
    up = {n: v * float(W * H) for n, v in up.items()}
    return cam, sc, up


def compare(name, cam, sc, up, semantic, threads=0, with_backward=True):
    kw = dict(colors_precomp=sc["colors_precomp"], scales=sc["scales"], rotations=sc["rotations"])
    if semantic:
        kw["semantics_precomp"] = sc["semantics_precomp"]
    res = {"scene": name}
    outs, sts, grads = {}, {}, {}
    for prec in ("f32", "fma"):
        t0 = time.time()
        outs[prec], sts[prec] = O.forward(cam, sc["means3D"], sc["opacities"], threads=threads, precision=prec, **kw)
        if with_backward:
            g = dict(color=up["color"].numpy(), semantic=up["semantic"].numpy() if semantic else None, depth=up["depth"].numpy(),
                     median=up["median"].numpy(), opacity=up["opacity"].numpy())
            grads[prec] = O.backward(sts[prec], cam, sc["means3D"], g, threads=threads, median_rule="forward", **kw)
        res["seconds_" + prec] = round(time.time() - t0, 2)
    a, b = sts["f32"], sts["fma"]
    P = a.P
    ra, rb = outs["f32"]["radii"], outs["fma"]["radii"]
    vis = (ra > 0) | (rb > 0)
    res.update(P=P, visible=int((ra > 0).sum()), num_rendered=[int(a.R), int(b.R)])
    res["radii_differ"] = int((ra != rb).sum())
    res["visibility_differs"] = int(((ra > 0) != (rb > 0)).sum())
    ta, tb = a.field("tiles_touched"), b.field("tiles_touched")
    res["tiles_touched_differ"] = int((ta != tb).sum())
    da, db = a.field("depths").view(np.uint32), b.field("depths").view(np.uint32)
    res["depth_bits_differ"] = int(((da != db) & vis).sum())
    ulp = np.abs(da.astype(np.int64) - db.astype(np.int64))[vis]
    res["depth_max_ulps"] = int(ulp.max()) if ulp.size else 0
    m2a, m2b = a.field("means2D"), b.field("means2D")
    res["means2D_bits_differ"] = int((np.any(m2a.view(np.uint32) != m2b.view(np.uint32), axis=1) & vis).sum())
    ca, cb = a.field("conic_opacity"), b.field("conic_opacity")
    res["conic_bits_differ"] = int((np.any(ca.view(np.uint32) != cb.view(np.uint32), axis=1) & vis).sum())
    # sorted lists: per tile, is the sequence of Gaussian ids the same?
    rga, rgb_ = a.field("ranges"), b.field("ranges")
    res["tile_ranges_differ"] = int(np.any(rga != rgb_, axis=1).sum())
    va, vb = a.field("vals"), b.field("vals")
    T = rga.shape[0]
    order_diff = 0
    members_diff = 0
    for t in range(T):
        la, lb = va[rga[t, 0]:rga[t, 1]], vb[rgb_[t, 0]:rgb_[t, 1]]
        if la.shape != lb.shape or not np.array_equal(la, lb):
            order_diff += 1
            if la.shape != lb.shape or not np.array_equal(np.sort(la), np.sort(lb)):
                members_diff += 1
    res["tiles"] = T
    res["tiles_whose_sorted_list_differs"] = order_diff
    res["tiles_whose_membership_differs"] = members_diff
    if a.R == b.R:
        ka, kb = a.field("keys"), b.field("keys")
        res["sorted_keys_differ"] = int((ka != kb).sum())
    else:
        res["sorted_keys_differ"] = None   # lists of different length
    na, nb = a.field("n_contrib"), b.field("n_contrib")
    res["n_contrib_differ_pixels"] = int((na != nb).sum())
    res["pixels"] = int(na.size)
    img = {}
    for n in ("color", "depth", "median_depth", "opacity") + (("semantic",) if semantic else ("mask",)):
        x, y = outs["f32"][n].astype(np.float64), outs["fma"][n].astype(np.float64)
        img[n] = float(np.abs(x - y).max() / max(np.abs(x).max(), 1e-30))
    res["image_max_abs_diff_over_max"] = img
    if with_backward:
        gd = {}
        for n in ("means3D", "means2D", "opacities", "colors_precomp", "scales", "rotations") + (("semantics_precomp",) if semantic else ()):
            x, y = grads["f32"][n].astype(np.float64), grads["fma"][n].astype(np.float64)
            gd[n] = float(np.abs(x - y).max() / max(np.abs(x).max(), 1e-30))
        res["gradient_max_abs_diff_over_max"] = gd
    for s in sts.values():
        s.free()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", nargs="+", default=["goldens", "small"], choices=["goldens", "small", "headline", "headline_tilted", "stress"])
    ap.add_argument("--json", default=None)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    out = []
    todo = {}
    if "goldens" in args.scenes:
        todo.update(GOLDEN)
    if "small" in args.scenes:
        todo.update(SMALL)
    for name, (W, H, P, K, kind, sm, semantic, bg, seed, behind) in todo.items():
        cam, sc, up = scenes.build(W, H, P, K, seed=seed, kind=kind, scale_mult=sm, bg=bg, behind_frac=behind)
        out.append(compare(name, cam, sc, up, semantic, args.threads))
        print(json.dumps(out[-1]), flush=True)
    if "headline" in args.scenes:
        cam, sc, up = bench_scene(1200, 680, 500000, 26, "slam")
        out.append(compare("headline 1200x680 P=500000 K=26 (bench.py scene)", cam, sc, up, True, args.threads))
        print(json.dumps(out[-1]), flush=True)
    if "headline_tilted" in args.scenes:
        cam, sc, up = bench_scene(1200, 680, 500000, 26, "slam", rank=3)
        out.append(compare("headline size, camera of rank 3's keyframe (rotated 0.03 rad, translated)", cam, sc, up, True, args.threads))
        print(json.dumps(out[-1]), flush=True)
    if "stress" in args.scenes:
        cam, sc, up = bench_scene(1920, 1080, 2000000, 74, "slam")
        out.append(compare("stress 1920x1080 P=2000000 K=74 (bench.py scene)", cam, sc, up, True, args.threads))
        print(json.dumps(out[-1]), flush=True)
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
