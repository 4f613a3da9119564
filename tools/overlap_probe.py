#!/usr/bin/env python3
"""Does a bandwidth-bound fill on a SIDE stream hide behind the forward's latency-bound front end?  Times the bench step (a) as is and
(b) with an extra memset of the backward's accumulation rows issued on a second stream at the start of the forward and joined before the
backward.  If (b) - (a) is ~0 the backward's own zero pass (bwd_zero stage) can move there.
    python tools/overlap_probe.py [--P 500000 --K 26 --width 1200 --height 680]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hier-slam_amd"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=500000)
    ap.add_argument("--K", type=int, default=26)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=680)
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    wl = bench.Workload(dev, a.width, a.height, a.P, a.K, "slam")
    row_floats = 16 * ((a.K + 22 + 15) // 16)
    fill = torch.empty(a.P * row_floats, dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(dev)

    def plain():
        wl.step()

    def with_fill():
        main_s = torch.cuda.current_stream(dev)
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            fill.zero_()
        done = torch.cuda.Event()
        done.record(side)
        # the forward and the backward both run inside step(); joining before the whole step's end is the conservative emulation
        wl.step_forward_then(lambda: main_s.wait_event(done))

    def step_forward_then(hook):
        leaf = wl.leaf
        means2D = torch.zeros(wl.P, 3, device=dev, requires_grad=True)
        out = wl.renderer(means3D=leaf["means3D"], means2D=means2D, opacities=leaf["opacities"], colors_precomp=leaf["colors_precomp"],
                          scales=leaf["scales"], rotations=leaf["rotations"], semantics_precomp=leaf["semantics_precomp"])
        color, radii, sem, depth, median, opac = out
        for n in wl.names:
            leaf[n].grad = None
        hook()
        torch.autograd.backward([color, sem, depth, median, opac], wl.upd)
    wl.step_forward_then = step_forward_then

    def timeit(fn):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(a.steps):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / a.steps * 1e3)
        return best
    r = {"P": a.P, "K": a.K, "fill_MB": fill.numel() * 4 / 1e6}
    r["plain_ms"] = timeit(plain)
    r["with_side_fill_ms"] = timeit(with_fill)
    r["plain_again_ms"] = timeit(plain)
    r["fill_alone_ms"] = timeit(lambda: fill.zero_())
    print(json.dumps(r))


if __name__ == "__main__":
    main()
