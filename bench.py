#!/usr/bin/env python
"""bench.py — forward+backward renders/s of the MI355X-native Hier-SLAM rasterizer.

One "step" = one GaussianRasterizer_semantic forward + one backward with dense upstream gradients on
all five differentiable outputs (colour, K semantic logits, depth, median depth, final opacity), on the
workload BASELINE.json's metric is quoted on: synthetic 1200x680 frame, 500k SLAM-like Gaussians, K=26
(the reference's default tree, config.h:18).  Inputs are resident in HBM before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank renders a different keyframe
of the same Gaussian set and the per-Gaussian gradients are summed with one bucketed all-reduce over
RCCL — the keyframe-parallel mapping step of SURVEY.md §8e.  value = renders of all ranks / max time.

Prints ONE JSON line (rank 0) carrying `roofline` (dominant kernel, timed live with HIP events on the
launch stream through the library's hsr_profile hooks) and `cpu_baseline` (the oracle on the host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "hier-slam_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(P, V, R, W, H, K, e=0):
    """Per-stage algorithmic HBM bytes of one render (SURVEY.md §8d / BASELINE.md §2), from measured V, R."""
    N = W * H
    T = ((W + 15) // 16) * ((H + 15) // 16)
    bit = max(1, int(np.ceil(np.log2(T)))) if T > 1 else 1
    if (1 << bit) == T:
        bit += 1  # getHigherMsb(T) of an exact power of two is log2(T)+1
    s = -(-(32 + bit) // 8)
    return {
        "fwd_preprocess": 44 * P + 8 * P + 52 * V,
        "fwd_scan": 8 * P,
        "fwd_duplicate": 20 * P + 12 * R,
        "fwd_sort": 24 * s * R,
        "fwd_ranges": 8 * R + 8 * T,
        "fwd_render": R * (44 + 4 * K) + 4 * N * (K + 8),
        "bwd_zero": 4 * P * (K + 28),
        "bwd_render": R * (44 + 4 * K * e) + 4 * N * (K + 8) + 8 * R * (K + 10),
        "bwd_preprocess": 164 * V + 4 * P,
    }


class _Profile(C.Structure):
    _fields_ = [("ms", C.c_double * 9), ("calls", C.c_uint64 * 9)]


def perturbed_w2c(rank):
    """rank 0: identity; other ranks: a small SE(3) perturbation (a different keyframe of the window)"""
    w2c = np.eye(4)
    if rank:
        a = 0.01 * rank
        w2c[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        w2c[:3, 3] = [0.01 * rank, -0.005 * rank, 0.0]
    return w2c


def cpu_baseline(args, sc, cam_cpu, up):
    """the oracle (a plain-C port of the reference's algorithm; the reference has no CPU renderer) on the host cores"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(threads, int(os.environ.get("HSR_CPU_THREADS", "64")))  # OpenMP scaling of the oracle flattens out
    kw = dict(colors_precomp=sc["colors_precomp"], semantics_precomp=sc["semantics_precomp"], scales=sc["scales"],
              rotations=sc["rotations"])
    g = {n: v.numpy() for n, v in up.items()}
    n_done, t0 = 0, time.time()
    while True:
        out, st = O.forward(cam_cpu, sc["means3D"], sc["opacities"], threads=threads, **kw)
        O.backward(st, cam_cpu, sc["means3D"], g, threads=threads, **kw)
        st.free()
        n_done += 1
        el = time.time() - t0
        if el > args.cpu_seconds or n_done >= 50:
            break
    return {"value": n_done / el, "unit": "renders/s", "cores": threads, "kind": "port",
            "sample": "%d fwd+bwd renders of the same %dx%d / %d Gaussians / K=%d workload by the OpenMP C oracle "
                      "(oracle/hsr_oracle.c) on all %d host threads" % (n_done, args.width, args.height, args.P, args.K, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 0.13 s timed at the headline sizes: long enough to average out host hiccups
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--P", type=int, default=500000)
    ap.add_argument("--K", type=int, default=26)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=680)
    ap.add_argument("--kind", default="slam", choices=["slam", "aniso"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bound on the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pin", action="store_true", help="leave the process free to migrate over all CPUs")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket stages with HIP events")
    args = ap.parse_args()
    # Keep the process — main thread, autograd engine thread, HIP runtime threads — on the CPUs that share an L3 with the one
    # it started on (numactl-style; the full set is restored for the CPU-baseline leg).  With the threads that hand work to
    # each other under one L3 the host side of a step takes ~0.13 instead of ~0.22 ms on a 2-socket EPYC box, which is slack the
    # 0.62 ms device step needs on a busy host.
    libc_cpu, full_affinity = -1, None
    try:
        import ctypes as _ct
        libc_cpu = int(_ct.CDLL(None).sched_getcpu())
        if not args.no_pin:
            full_affinity = os.sched_getaffinity(0)

            def l3_of(cpu):
                with open("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % cpu) as f:
                    out = set()
                    for part in f.read().strip().split(","):
                        a, _, b = part.partition("-")
                        out.update(range(int(a), int(b or a) + 1))
                return frozenset(out & full_affinity)
            mine = l3_of(libc_cpu)
            lr = int(os.environ.get("LOCAL_RANK", "0"))
            if lr:   # one L3 domain per rank: the lr-th one after the domain this rank started on
                groups = []
                for c in sorted(full_affinity):
                    g = l3_of(c)
                    if g not in groups:
                        groups.append(g)
                mine = groups[(groups.index(mine) + lr) % len(groups)]
            if len(mine) >= 4:
                os.sched_setaffinity(0, mine)
            else:
                full_affinity = None
    except Exception:
        full_affinity = None
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # rehearsal knobs for a 1-GPU box (never set by the driver): all ranks on device 0, gloo instead of RCCL
    share_gpu = os.environ.get("HSR_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("HSR_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", 0 if share_gpu else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer_semantic, _C
    from hsr_utils.camera import replica_intrinsics, setup_camera_tensors
    from hsr_utils.parallel import GradientBucket
    from hsr_utils.synthetic import make_scene, make_upstream_grads

    W, H, K, P = args.width, args.height, args.K, args.P
    kmat = replica_intrinsics(W, H)
    cam_cpu = setup_camera_tensors(W, H, kmat, perturbed_w2c(rank))
    cam = GaussianRasterizationSettings(**{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in cam_cpu.items()})
    sc = make_scene(P, W, H, K, kmat, seed=0, kind=args.kind)  # same Gaussians on every rank (replicated parameters)
    up = make_upstream_grads(W, H, K, seed=1 + rank)
    names = ("means3D", "colors_precomp", "semantics_precomp", "opacities", "scales", "rotations")
    leaf = {n: sc[n].to(dev).requires_grad_(True) for n in names}
    upd = [up[n].to(dev) for n in ("color", "semantic", "depth", "median", "opacity")]
    renderer = GaussianRasterizer_semantic(cam)
    # N > 1: the one exchange step of the sharded path — the per-Gaussian gradients of the ranks' keyframes are
    # summed with one bucketed all-reduce per step (76 MB at the headline sizes).  Two buckets in flight: the all-reduce of step
    # i runs on RCCL's stream while step i + 1 renders, and a bucket is waited for only when it is packed again (the timed
    # region ends with both drained).  As at N = 1 there is no optimizer inside a step.
    buckets = [GradientBucket([leaf[n].shape for n in names], dev) for _ in range(2)] if world > 1 else None
    works = [None, None]
    info = {"i": 0}

    def step():
        means2D = torch.zeros(P, 3, device=dev, requires_grad=True)  # hierslam.py:895 retains this grad for densification
        color, radii, sem, depth, median, opac = renderer(
            means3D=leaf["means3D"], means2D=means2D, opacities=leaf["opacities"], colors_precomp=leaf["colors_precomp"],
            scales=leaf["scales"], rotations=leaf["rotations"], semantics_precomp=leaf["semantics_precomp"])
        info["R"] = color.grad_fn.num_rendered
        info["radii"] = radii
        for n in names:
            leaf[n].grad = None
        torch.autograd.backward([color, sem, depth, median, opac], upd)
        if buckets is not None:
            b = info["i"] & 1
            info["i"] += 1
            if works[b] is not None:
                works[b].wait()
            buckets[b].pack([leaf[n].grad for n in names])
            works[b] = buckets[b].all_reduce(async_op=True)

    def sync():
        for b in range(2):
            if works[b] is not None:
                works[b].wait()
                works[b] = None
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    # Stage timers cost two event records per stage and call, and eight timed stages per step slow the launch sequence
    # down by ~8 %.  So: (1) an UNTIMED instrumented pre-pass gives the full stage table and names the dominant stage;
    # (2) during the timed region only that stage is bracketed by HIP events (on the launch stream, inside the library).
    prof = prof_all = None
    dom_stage = -1
    if not args.no_profile:
        _C._lib.hsr_profile_read(None, 1)
        _C._lib.hsr_profile_select(0xFFFFFFFF)
        _C._lib.hsr_profile_enable(1)
        n_pre = max(3, min(10, args.warmup))
        for _ in range(n_pre):
            step()
        sync()
        prof_all = _Profile()
        _C._lib.hsr_profile_read(C.byref(prof_all), 1)
        dom_stage = max(range(9), key=lambda i: prof_all.ms[i])
        _C._lib.hsr_profile_select(1 << dom_stage)
    _C._lib.hsr_profile_host_wait_ms(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_wait_ms = float(_C._lib.hsr_profile_host_wait_ms(1)) / args.steps
    sync()
    t1 = time.perf_counter()
    if not args.no_profile:
        prof = _Profile()
        _C._lib.hsr_profile_enable(0)
        _C._lib.hsr_profile_read(C.byref(prof), 1)
        _C._lib.hsr_profile_select(0xFFFFFFFF)
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        R = int(info["R"])
        V = int((info["radii"] > 0).sum().item())
        ms_per_step = 1e3 * elapsed / args.steps
        out = {
            "metric": "fwd+bwd renders/sec @1200x680, 500k Gaussians, 4-level tree; grad max-abs-err vs ref",
            "value": world * args.steps / elapsed, "unit": "renders/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "semantic fwd+bwd render, %dx%d, P=%d %s Gaussians, K=%d semantic channels, "
                                   "dense upstream grads on colour/semantic/depth/median/opacity" % (W, H, P, args.kind, K),
                       "P": P, "visible": V, "num_rendered": R, "width": W, "height": H, "K": K,
                       "parallelism": ("keyframe-parallel x%d: one keyframe per rank per step, 2 gradient buckets in flight (all-reduce of step i overlaps render i+1)" % world) if world > 1 else "single GPU",
                       "api": "diff_gaussian_rasterization.GaussianRasterizer_semantic (torch autograd) -> C ABI"},
        }
        if prof is not None:
            alg = algorithmic_bytes(P, V, R, W, H, K)
            stages = {}
            for i in range(9):
                nm = _C._lib.hsr_stage_name(i)
                nm = nm.decode() if isinstance(nm, bytes) else C.cast(nm, C.c_char_p).value.decode()
                # per STEP (a stage can be several timed sections per step); dominant stage: from the timed region
                src, nst = (prof, args.steps) if (i == dom_stage and prof.calls[i]) else (prof_all, n_pre)
                if src.calls[i]:
                    stages[nm] = {"ms": src.ms[i] / nst, "alg_bytes": alg[nm], "GBps": alg[nm] / (src.ms[i] / nst * 1e-3) / 1e9}
            dom = max(stages, key=lambda n: stages[n]["ms"])
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": stages[dom]["GBps"] / HBM_PEAK_GBS, "traffic": None,
                               "kernel_ms": stages[dom]["ms"], "alg_bytes_per_launch": stages[dom]["alg_bytes"]}
            # HBM traffic of the dominant kernel from the PMC passes of tools/profile_gpu.sh (committed under
            # profiles/; counters cannot be read from inside this process), when it was taken on this workload
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
                wl = tj["workload"]
                if (wl["P"], wl["width"], wl["height"], wl["K"], wl["kind"]) == (P, W, H, K, args.kind):
                    out["roofline"]["traffic"] = tj["traffic_bytes_per_launch"].get(dom)
                    out["roofline"]["traffic_source"] = tj["source"]
            except Exception:
                pass
            out["stages_ms"] = {n: round(v["ms"], 4) for n, v in stages.items()}
            out["stages_note"] = ("HIP events inside the library on the launch stream; the roofline kernel (%s) is timed during the "
                                  "timed region, the other stages in an untimed instrumented pre-pass (timing all eight stages "
                                  "costs ~8 %% of the step)" % dom)
            tot_alg = sum(alg.values())
            out["whole_render"] = {"alg_bytes": tot_alg, "GBps": tot_alg / (ms_per_step * 1e-3) / 1e9,
                                   "frac_of_hbm_peak": tot_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "device_ms_sum": round(sum(v["ms"] for v in stages.values()), 4)}
        # how long the host sat blocked on the device per step (the forward's num_rendered read-back): about one device step =
        # device-bound; near zero while ms_per_step exceeds the device time = this box's host cannot keep the device fed
        out["host"] = {"blocked_on_device_ms_per_step": round(host_wait_ms, 4), "cpu_at_start": int(libc_cpu), "pinned_to_l3_cpus": (len(os.sched_getaffinity(0)) if full_affinity else 0),
                       "note": "ms_per_step - blocked = host-side work per step (Python glue + launches)"}
        if world == 1 and not args.no_cpu_baseline:
            if full_affinity:
                os.sched_setaffinity(0, full_affinity)   # the oracle gets every host thread it is allowed
            out["cpu_baseline"] = cpu_baseline(args, sc, cam_cpu, up)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
