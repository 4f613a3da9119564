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
launch stream through the library's hsr_profile hooks), `cpu_baseline` (the oracle on the host cores) and
`parity` — the metric's second half, "grad max-abs-err vs ref": one more GPU step compared with the oracle
render of the SAME workload that the cpu_baseline leg computes anyway (bit-equality of the integer state,
image and gradient errors).

`python bench.py --gpus N` without a launcher starts the N ranks itself (a child `python -m
torch.distributed.run`, before this process has touched the GPU) and relays rank 0's line.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
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


def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, sc, cam_cpu, up):
    """the oracle (a plain-C port of the reference's algorithm; the reference has no CPU renderer) on the host cores.
    Returns (cpu_baseline dict, (outputs, gradients, state) of the FIRST oracle render — the parity block's expectation)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    kw = dict(colors_precomp=sc["colors_precomp"], semantics_precomp=sc["semantics_precomp"], scales=sc["scales"],
              rotations=sc["rotations"])
    g = {n: v.numpy() for n, v in up.items()}
    # the thread count is chosen by measurement, not assumed (round 3 capped it at 64 of a 256-thread box): one untimed fwd+bwd render at
    # 64 / 128 / all available threads (HSR_CPU_THREADS pins it), the fastest count runs the bounded sample
    cands = [int(os.environ["HSR_CPU_THREADS"])] if os.environ.get("HSR_CPU_THREADS") else sorted({min(avail, c) for c in (64, 128, avail)})
    probe = {}
    if len(cands) > 1 and args.cpu_seconds > 0:
        for c in cands:
            tp = time.time()
            o_, s_ = O.forward(cam_cpu, sc["means3D"], sc["opacities"], threads=c, **kw)
            O.backward(s_, cam_cpu, sc["means3D"], g, threads=c, median_rule="forward", **kw)
            s_.free()
            probe[c] = time.time() - tp
        threads = min(probe, key=probe.get)
    else:
        threads = cands[-1]
    n_done, t0, first = 0, time.time(), None
    while True:
        out, st = O.forward(cam_cpu, sc["means3D"], sc["opacities"], threads=threads, **kw)
        # dL_dmedian_depth to the splat the forward recorded, as the product does (the reference's reconstructed-T rule differs
        # only on rounding ties; the count of such pixels is reported in the parity block)
        # the first render also evaluates the oracle's tie bounds (every threshold decision taken within ulps, the other way): what the
        # parity block may allow on exactly those pixels / gradient rows
        gr = O.backward(st, cam_cpu, sc["means3D"], g, threads=threads, median_rule="forward", bounds=first is None, **kw)
        st.median_rule_disagreements = gr["median_rule_disagreements"]
        st.grad_bounds, st.bounds_info = gr.get("bounds"), gr.get("bounds_info")
        n_done += 1
        el = time.time() - t0
        if first is None:
            first = (out, gr, st)
        else:
            st.free()
        if el > args.cpu_seconds or n_done >= 50:
            break
    return ({"value": n_done / el, "unit": "renders/s", "cores": threads, "kind": "port",
             "cpu_model": cpu_model_name(), "host_logical_cpus": os.cpu_count(), "cpus_available_to_the_process": avail,
             "threads_probed_seconds_per_render": {str(c): round(v, 3) for c, v in probe.items()},
             "sample": "%d fwd+bwd renders of the same %dx%d / %d Gaussians / K=%d workload by the OpenMP C oracle "
                       "(oracle/hsr_oracle.c) on %d host threads (the fastest of %s)" % (n_done, args.width, args.height, args.P, args.K, threads,
                                                                                        "/".join(str(c) for c in cands))},
            first)


def parity_block(cam_cpu, sc, up, oracle_first, dev, brief=False):
    """One more GPU step through the public API (tests/harness.run_gpu: forward, loss = sum(out * upstream), backward,
    state read-back) against the oracle render of the same inputs.  The oracle is the checker here, never the product."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness
    out_o, gr_o, st_o = oracle_first
    out_g, gr_g, st_g = harness.run_gpu(cam_cpu, sc, up, semantic=True, dev=str(dev))
    grads_o = {n: gr_o[n] for n in ("means3D", "opacities", "means2D", "scales", "rotations", "colors_precomp", "semantics_precomp")}
    rep = harness.parity_report(out_g, gr_g, st_g, out_o, grads_o, st_o, semantic=True)
    rep["reference"] = ("oracle/hsr_oracle.c, the builder's plain-C restatement of forward.cu / backward.cu / rasterizer_impl.cu "
                        "— PARITY UNPINNED: the reference holds no fixtures for this path and cannot be built here (DESIGN.md §2)")
    rep["tolerance"] = ("integers bit-exact; images and gradients 1e-4 of each tensor's largest entry (north star), and "
                        "element-wise 1e-4 * max(|exp_i|, f * max|exp|) with f = %g (%g for scales / rotations: ill-conditioned chain, "
                        "tests/harness.py); n_contrib / median depth depend on exp() last-ulp ties at the alpha >= 1/255 and "
                        "T < 0.5 thresholds and are reported as counts; dL_dmedian_depth goes to the splat the FORWARD recorded "
                        "(oracle median_rule 'forward'; oracle_median_rule_disagreements = pixels where the reference's "
                        "reconstructed-T rule would pick a neighbour)" % (harness.FLOOR_FRAC, harness.FLOOR_FRAC_COV))
    ok_int = all(rep[k] for k in ("num_rendered_equal", "radii_equal", "tiles_touched_equal", "keys_equal", "vals_equal", "ranges_equal"))
    rep["pass"] = bool(ok_int and max(rep["grad_err_over_max"].values()) <= 1e-4 and max(rep["image_err_over_max"].values()) <= 1e-4
                       and max(rep["grad_elementwise_err"].values()) <= 1e-4)
    # "pass" is strict on purpose.  A threshold decision taken the other way (alpha >= 1/255 within an ulp: v_exp_f32 vs glibc) moves a
    # pixel by a whole contribution and is likely somewhere in 2M pixels x hundreds of splats.  The oracle evaluates every decision it
    # took within ulps of a threshold BOTH ways and bounds the difference per pixel and per gradient entry; pass_with_tie_bounds adds
    # TIE_SLACK x that bound entry by entry (nothing is left out) and requires the rows it loosens to stay under TIE_LOOSENED_FRAC
    rows = max(1, rep["grad_rows"])
    rep["tie_bound_rule"] = ("|err_i| <= ordinary bound_i + %g x oracle tie bound_i; rows whose tie bound exceeds their ordinary bound "
                             "<= max(%d, %g x rows)" % (harness.TIE_SLACK, harness.TIE_LOOSENED_MIN, harness.TIE_LOOSENED_FRAC))
    rep["pass_with_tie_bounds"] = bool(ok_int and max(rep["grad_err_over_max_beyond_tie_bound"].values()) <= 1e-4
                                       and max(rep["image_err_over_max_beyond_tie_bound"].values()) <= 1e-4
                                       and max(rep["grad_elementwise_err_beyond_tie_bound"].values()) <= 1e-4
                                       and rep["oracle_tie_bounds"].get("overflow_pixels", 1) == 0
                                       and max(rep["grad_rows_loosened_by_tie_bound"].values()) <= max(harness.TIE_LOOSENED_MIN, harness.TIE_LOOSENED_FRAC * rows))
    st_o.free()
    if brief:   # an extra workload's entry: verdicts and the headline figures only
        keep = ("pass", "pass_with_tie_bounds", "num_rendered_equal", "radii_equal", "tiles_touched_equal", "keys_equal", "vals_equal", "ranges_equal",
                "n_contrib_mismatch", "median_pos_mismatch", "oracle_tie_risk_pixels", "oracle_tie_bounds", "image_err_over_max",
                "image_err_over_max_beyond_tie_bound", "grad_err_over_max", "grad_err_over_max_beyond_tie_bound", "grad_elementwise_err",
                "grad_elementwise_err_beyond_tie_bound", "grad_rows_loosened_by_tie_bound", "grad_rows", "tie_bound_rule", "reference")
        rep = {k: rep[k] for k in keep if k in rep}
    return rep


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n, argv, port, python=None):
    """the command `bench.py --gpus N` runs when no launcher started it: one rank per GPU of this node under
    torch.distributed.run (the same line the driver uses), rendezvous on 127.0.0.1"""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv):
    """Parent side of `python bench.py --gpus N` (N > 1, no WORLD_SIZE): nothing in this process has touched the GPU yet
    (torch is imported, no torch.cuda call made); the ranks run in a CHILD process tree — never an exec of this one —
    and rank 0's JSON line is relayed.  Returns the exit code."""
    cmd = launch_command(n, argv, free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    return proc.returncode if (proc.returncode or line is not None) else 1


def parse_cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            out.update(range(int(a), int(b or a) + 1))
    return out


def choose_cpus(local_rank, n_local, affinity, l3_of, gpu_local_cpus=None, start_cpu=None):
    """CPUs for one rank's process (main thread, autograd engine thread, HIP runtime threads): one L3 domain, chosen by a
    rule that does not depend on where the launcher happened to start the process.

    n_local == 1: the L3 domain of `start_cpu` (no migration).  n_local > 1: the L3 domains of the CPUs this job may use,
    in ascending CPU order, restricted to `gpu_local_cpus[r]` (the CPUs on the NUMA node of rank r's GPU) when known;
    the ranks that share a node take that node's domains at an even stride, so no two ranks share a domain while
    domains last.  Returns a frozenset, or None to leave the affinity alone (fewer than 4 CPUs in the domain)."""
    affinity = frozenset(affinity)
    if n_local <= 1:
        mine = frozenset(l3_of(start_cpu)) & affinity if start_cpu is not None else affinity
        return mine if len(mine) >= 4 else None

    def domains(cpus):
        seen, out = set(), []
        for c in sorted(cpus):
            if c in seen:
                continue
            d = frozenset(l3_of(c)) & affinity
            seen |= d
            if d:
                out.append(d)
        return out
    node_of = lambda r: frozenset(gpu_local_cpus[r]) & affinity if (gpu_local_cpus and gpu_local_cpus.get(r)) else affinity
    mine_node = node_of(local_rank) or affinity
    peers = [r for r in range(n_local) if (node_of(r) or affinity) == mine_node]
    doms = domains(mine_node)
    if not doms:
        return None
    pick = doms[(peers.index(local_rank) * len(doms)) // len(peers)] if len(doms) >= len(peers) else doms[peers.index(local_rank) % len(doms)]
    return pick if len(pick) >= 4 else None


def gpu_numa_cpus(n_local):
    """{local rank: CPUs local to that rank's GPU} from sysfs via the device's PCI address; {} when it cannot be read"""
    out = {}
    try:
        for r in range(n_local):
            pr = torch.cuda.get_device_properties(r)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            with open("/sys/bus/pci/devices/%s/local_cpulist" % bdf) as f:
                cpus = parse_cpulist(f.read())
            if cpus:
                out[r] = cpus
    except Exception:
        return {}
    return out


# the rest of BASELINE.json's workload list (north star: "100k/500k/2M Gaussians", K = 16 "4-level tree", the reference's large label
# sets config.h:18, the stress configuration): (width, height, P, K, kind, tag)
EXTRA_WORKLOADS = [
    (1200, 680, 100000, 26, "slam", "100k"),
    # the same with the opt-in non-blocking forward (diff_gaussian_rasterization.set_async_forward): at this size the blocking
    # read-back of num_rendered, not the device, sets the step time
    (1200, 680, 100000, 26, "slam", "100k, non-blocking forward (opt-in)"),
    (1200, 680, 500000, 26, "slam", "500k (headline), non-blocking forward (opt-in)"),
    (1200, 680, 300000, 26, "slam", "300k (BASELINE.json configs[1]: ~300k Gaussians)"),
    (1200, 680, 2000000, 26, "slam", "2M"),
    (1200, 680, 500000, 16, "slam", "K=16 (ScanNet NYU40 4-level tree)"),
    # the backward tile kernel without its K-dependent part: K = 0 with every gradient, and the geometry-only backward
    # of a TRACKING iteration (only the camera pose is optimised: scripts/hierslam.py:1683-1860 — the loop Hier-SLAM runs most often)
    (1200, 680, 500000, 0, "slam", "K=0 (RGB-D only, every gradient)"),
    (1200, 680, 500000, 26, "slam", "tracking iteration (geometry-only backward: gradients for the means alone)"),
    (1200, 680, 500000, 26, "aniso", "anisotropic"),
    (1200, 680, 500000, 74, "slam", "K=74 (ScanNet large tree)"),
    (1200, 680, 500000, 102, "slam", "K=102 (Replica flat labels)"),
    (1920, 1080, 2000000, 74, "slam", "stress"),
]


class Workload:
    """one synthetic workload resident on the device + the step the bench times on it (public autograd API, fwd + bwd)"""

    names = ("means3D", "colors_precomp", "semantics_precomp", "opacities", "scales", "rotations")

    def __init__(self, dev, W, H, P, K, kind, rank=0, world=1, device_tensors=True, behind_frac=0.0, geo=False, keyframes=1):
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer_semantic
        from hsr_utils.camera import replica_intrinsics, setup_camera_tensors
        from hsr_utils.synthetic import make_scene, make_upstream_grads
        self.dev, self.W, self.H, self.P, self.K, self.kind, self.world = dev, W, H, P, K, kind, world
        kmat = replica_intrinsics(W, H)
        self.cam_cpu = setup_camera_tensors(W, H, kmat, perturbed_w2c(rank))
        self.sc = make_scene(P, W, H, K, kmat, seed=0, kind=kind, behind_frac=behind_frac)  # same Gaussians on every rank (replicated parameters)
        self.behind_frac = behind_frac
        self.geo = geo
        self.up = make_upstream_grads(W, H, K, seed=1 + rank)
        self.exchange = None
        self.stale = False
        self.keyframes = max(1, int(keyframes))
        self.steps_done = 0
        self.info = {}
        if device_tensors:
            cam = GaussianRasterizationSettings(**{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in self.cam_cpu.items()})
            # N > 1: this rank's G keyframes of the mapping window (shard_keyframes: rank r renders keyframes r, r + N, ...): G cameras, G
            # sets of upstream gradients, the same replicated Gaussians
            self.kf = []
            for g in range(1, self.keyframes):
                cam_g = setup_camera_tensors(W, H, kmat, perturbed_w2c(rank + g * world))
                up_g = make_upstream_grads(W, H, K, seed=1 + rank + g * world)
                self.kf.append((GaussianRasterizer_semantic(GaussianRasterizationSettings(**{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in cam_g.items()})),
                                [up_g[n].to(dev) for n in ("color", "semantic", "depth", "median", "opacity")]))
            # geo: a tracking iteration (scripts/hierslam.py:1683-1860) — only the camera pose is optimised, so autograd asks for the
            # gradient of the (transformed) means alone and the backward takes its geometry-only path
            self.leaf = {n: self.sc[n].to(dev).requires_grad_(not geo or n == "means3D") for n in self.names}
            self.upd = [self.up[n].to(dev) for n in ("color", "semantic", "depth", "median", "opacity")]
            self.renderer = GaussianRasterizer_semantic(cam)

    def describe(self):
        return ("semantic fwd+bwd render, %dx%d, P=%d %s Gaussians%s, K=%d semantic channels, dense upstream grads on "
                "colour/semantic/depth/median/opacity%s" % (self.W, self.H, self.P, self.kind,
                                                            (" (%.0f %% behind the camera)" % (100 * self.behind_frac)) if self.behind_frac else "", self.K,
                                                            "; gradients for means3D / means2D only (tracking iteration)" if self.geo else ""))

    def release(self):
        self.leaf = self.upd = self.renderer = None
        torch.cuda.empty_cache()

    def step(self):
        """N = 1: one render (forward + backward).  N > 1: one optimizer step's worth of the mapping loop (scripts/hierslam.py:1966-2057) —
        this rank's G keyframes rendered one after the other, their gradients accumulated in the exchange bucket, ONE exchange, and
        (unless --stale-gradients) the wait for the sum: the point at which an Adam step could run."""
        leaf, dev = self.leaf, self.dev
        if self.exchange is not None:
            self.exchange.begin_step()   # the backwards write / accumulate their gradient outputs into the bucket
        else:
            for n in self.names:
                leaf[n].grad = None
        for g in range(self.keyframes):
            renderer, upd = (self.renderer, self.upd) if g == 0 else self.kf[g - 1]
            means2D = torch.zeros(self.P, 3, device=dev, requires_grad=True)  # hierslam.py:895 retains this grad for densification
            color, radii, sem, depth, median, opac = renderer(
                means3D=leaf["means3D"], means2D=means2D, opacities=leaf["opacities"], colors_precomp=leaf["colors_precomp"],
                scales=leaf["scales"], rotations=leaf["rotations"], semantics_precomp=leaf["semantics_precomp"])
            if g == 0:
                self.info["node"] = color.grad_fn
                self.info["radii"] = radii
            if self.exchange is not None:
                self.exchange.add_keyframe(radii)
            torch.autograd.backward([color, sem, depth, median, opac], upd)
        if self.exchange is not None:
            self.exchange.submit()
            if not self.stale:
                self.exchange.reduced(self.steps_done)   # the sum over all N x G keyframes has arrived
        self.steps_done += 1

    def sync(self):
        if self.exchange is not None:
            self.exchange.drain()
        torch.cuda.synchronize(self.dev)
        if self.world > 1:
            dist.barrier()
            torch.cuda.synchronize(self.dev)

    def run(self, steps, warmup, profile=True, lib=None):
        for _ in range(warmup):
            self.step()
        self.sync()
        # Stage timers cost two event records per stage and call, and eight timed stages per step slow the launch sequence
        # down by ~8 %.  So: (1) an UNTIMED instrumented pre-pass gives the full stage table and names the dominant stage;
        # (2) during the timed region only that stage is bracketed by HIP events (on the launch stream, inside the library).
        prof = prof_all = None
        dom_stage, n_pre = -1, 0
        if profile:
            lib.hsr_profile_read(None, 1)
            lib.hsr_profile_select(0xFFFFFFFF)
            lib.hsr_profile_enable(1)
            n_pre = max(3, min(10, warmup))
            for _ in range(n_pre):
                self.step()
            self.sync()
            prof_all = _Profile()
            lib.hsr_profile_read(C.byref(prof_all), 1)
            dom_stage = max(range(9), key=lambda i: prof_all.ms[i])
            lib.hsr_profile_select(1 << dom_stage)
        lib.hsr_profile_host_wait_ms(1)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        host_wait_ms = float(lib.hsr_profile_host_wait_ms(1)) / steps
        self.sync()
        t1 = time.perf_counter()
        if profile:
            prof = _Profile()
            lib.hsr_profile_enable(0)
            lib.hsr_profile_read(C.byref(prof), 1)
            lib.hsr_profile_select(0xFFFFFFFF)
        R = int(self.info["node"].num_rendered)
        V = int((self.info["radii"] > 0).sum().item())
        res = {"elapsed": t1 - t0, "steps": steps, "host_wait_ms": host_wait_ms, "R": R, "V": V}
        if prof is not None:
            alg = algorithmic_bytes(self.P, V, R, self.W, self.H, self.K)
            stages = {}
            for i in range(9):
                nm = lib.hsr_stage_name(i)
                nm = nm.decode() if isinstance(nm, bytes) else C.cast(nm, C.c_char_p).value.decode()
                # per STEP (a stage can be several timed sections per step); dominant stage: from the timed region
                src, nst = (prof, steps) if (i == dom_stage and prof.calls[i]) else (prof_all, n_pre)
                if src.calls[i]:
                    stages[nm] = {"ms": src.ms[i] / nst, "alg_bytes": alg[nm], "GBps": alg[nm] / (src.ms[i] / nst * 1e-3) / 1e9}
            res["stages"] = stages
            res["dominant"] = max(stages, key=lambda n: stages[n]["ms"])
        return res


def roofline_block(stages, dom, P, W, H, K, kind):
    """the dominant kernel against the HBM roofline the north star prescribes, plus what the committed PMC passes say bounds it"""
    rl = {"bound": "hbm", "kernel": dom, "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS,
          "unit": "GB/s", "frac": stages[dom]["GBps"] / HBM_PEAK_GBS, "traffic": None,
          "kernel_ms": stages[dom]["ms"], "alg_bytes_per_launch": stages[dom]["alg_bytes"],
          "bound_note": "`bound` names the roofline `peak` and `frac` refer to (the north star prescribes HBM; the contract's vocabulary is "
                        "hbm|mfma).  What actually limits the kernel, from the counters: `bound_by_counters`, `issue_frac`, `atomic_floor_ms`"}
    # HBM traffic of the dominant kernel from the PMC passes of tools/profile_gpu.sh (committed under
    # profiles/; counters cannot be read from inside this process), when it was taken on this workload
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        wl = tj["workload"]
        if (wl["P"], wl["width"], wl["height"], wl["K"], wl["kind"]) == (P, W, H, K, kind):
            rl["traffic"] = tj["traffic_bytes_per_launch"].get(dom)
            rl["traffic_source"] = tj["source"]
            if tj.get("limiter", {}).get(dom):   # what the PMC passes say bounds this kernel (the yardstick stays HBM)
                rl["limiter"] = tj["limiter"][dom]
            for key in ("bound_by_counters", "issue_frac", "atomic_floor_ms", "atomic_floor_frac_of_kernel", "hbm_frac_by_traffic"):
                if tj.get(key, {}).get(dom) is not None:
                    rl[key] = tj[key][dom]
    except Exception:
        pass
    return rl


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
    ap.add_argument("--geo", action="store_true", help="tracking iteration: gradients for means3D / means2D only (geometry-only backward)")
    ap.add_argument("--behind-frac", type=float, default=0.0, help="fraction of the Gaussians placed behind the camera (culled): a camera that sees part of the map")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bound on the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle leg (also skips the parity block)")
    ap.add_argument("--no-parity", action="store_true", help="time the oracle but do not compare the GPU step with it")
    ap.add_argument("--no-pin", action="store_true", help="leave the process free to migrate over all CPUs")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket stages with HIP events")
    ap.add_argument("--no-workloads", action="store_true", help="N = 1: only the headline workload, not the rest of the north star's list")
    ap.add_argument("--workload-steps", type=int, default=20, help="timed steps of each extra workload (at least 20)")
    ap.add_argument("--dense-exchange", action="store_true", help="N > 1: all-reduce every gradient row instead of the visible union")
    ap.add_argument("--keyframes-per-rank", type=int, default=0,
                    help="N > 1: keyframes each rank renders per optimizer step (gradients accumulated locally, ONE exchange per step); "
                         "default 3 = the Replica configuration's mapping window of 24 keyframes over 8 ranks")
    ap.add_argument("--stale-gradients", action="store_true",
                    help="N > 1: two buckets, the exchange of step i overlaps the renders of step i + 1 — only valid for a caller that applies "
                         "step i's gradients after rendering step i + 1 (the reference's loop does not: an Adam step sits between iterations)")
    ap.add_argument("--async-forward", action="store_true", help="opt-in non-blocking forward for the main workload too (diff_gaussian_rasterization.set_async_forward)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    # Keep the process — main thread, autograd engine thread, HIP runtime threads — on the CPUs that share an L3 with the one
    # it started on (numactl-style; the full set is restored for the CPU-baseline leg).  With the threads that hand work to
    # each other under one L3 the host side of a step takes ~0.13 instead of ~0.22 ms on a 2-socket EPYC box, which is slack the
    # 0.62 ms device step needs on a busy host.
    libc_cpu, full_affinity, pinned = -1, None, None
    n_local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    try:
        libc_cpu = int(C.CDLL(None).sched_getcpu())
        if not args.no_pin:
            full_affinity = os.sched_getaffinity(0)

            def l3_of(cpu):
                with open("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % cpu) as f:
                    return parse_cpulist(f.read())
            lr = int(os.environ.get("LOCAL_RANK", "0"))
            pinned = choose_cpus(lr, n_local, full_affinity, l3_of, gpu_numa_cpus(n_local) if n_local > 1 else None, libc_cpu)
            if pinned:
                os.sched_setaffinity(0, pinned)
            else:
                full_affinity = None
    except Exception:
        full_affinity = pinned = None
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
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it as `python bench.py --gpus %d` (it starts its "
                         "own ranks) or under torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus, args.gpus))
    if world > 1:
        assert dist.get_world_size() == world and (backend != "nccl" or dist.get_backend() == "nccl")

    from diff_gaussian_rasterization import _C
    from hsr_utils.parallel import GradientExchange

    W, H, K, P = args.width, args.height, args.K, args.P
    G = 1 if world == 1 else (args.keyframes_per_rank if args.keyframes_per_rank > 0 else 3)
    wl = Workload(dev, W, H, P, K, args.kind, rank, world, behind_frac=args.behind_frac, geo=args.geo, keyframes=G)
    exch = None
    if world > 1:
        # N > 1: the one exchange step of the sharded path, as a mapping step needs it: every rank renders its G keyframes
        # of the window, the gradients accumulate in the exchange bucket (the leaves' .grad tensors are VIEWS of it: no pack copy), ONE
        # exchange per optimizer step — only the rows of Gaussians some keyframe of some rank saw (visibility-sparse, bit-identical to the
        # dense sum: hsr_utils/parallel.py) — and the step ends when the sum has arrived.  --stale-gradients: two buckets, the exchange of
        # step i overlaps the renders of step i + 1 (round 3's default; not what the reference's loop allows).  As at N = 1 there is no
        # optimizer inside a step.
        exch = GradientExchange({"raster." + n: wl.leaf[n] for n in wl.names}, dev, depth=2 if args.stale_gradients else 1,
                                sparse=not args.dense_exchange)
        wl.stale = bool(args.stale_gradients)
    wl.exchange = exch
    if args.async_forward:
        import diff_gaussian_rasterization as dgr
        dgr.set_async_forward(True)
    res = wl.run(args.steps, args.warmup, profile=not args.no_profile, lib=_C._lib)
    if args.async_forward:
        dgr.set_async_forward(False)
    elapsed = res["elapsed"]
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    rank_cpus = None
    if world > 1:
        rank_cpus = [None] * world
        dist.all_gather_object(rank_cpus, sorted(os.sched_getaffinity(0)))
    if rank == 0:
        R, V = res["R"], res["V"]
        ms_per_step = 1e3 * elapsed / args.steps
        out = {
            "metric": "fwd+bwd renders/sec @1200x680, 500k Gaussians, 4-level tree; grad max-abs-err vs ref",
            "value": world * G * args.steps / elapsed, "unit": "renders/s", "n_gpus": (dist.get_world_size() if world > 1 else 1), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.describe(), "P": P, "visible": V, "num_rendered": R, "width": W, "height": H, "K": K,
                       "parallelism": ("keyframe-parallel x%d: a step = one optimizer step's worth of a mapping window of %d keyframes, %d per rank, "
                                       "gradients accumulated in the exchange bucket (views, no pack copy), ONE %s all-reduce per step, %s"
                                       % (world, world * G, G, "visibility-sparse" if not args.dense_exchange else "dense",
                                          "2 buckets in flight: the exchange of step i overlaps the renders of step i + 1 (--stale-gradients)"
                                          if args.stale_gradients else "the step ends when the sum has arrived (no overlap with the next step)"))
                                      if world > 1 else "single GPU",
                       "keyframes_per_rank": G, "renders_per_step": world * G,
                       "exchange_mode": (None if world == 1 else ("stale-gradients (pipelined)" if args.stale_gradients else "accumulate-then-exchange")),
                       "api": "diff_gaussian_rasterization.GaussianRasterizer_semantic (torch autograd) -> C ABI"
                              + (", non-blocking forward (opt-in)" if args.async_forward else "")},
        }
        if exch is not None:
            out["exchange"] = exch.stats()
        if res.get("stages"):
            stages, dom = res["stages"], res["dominant"]
            out["roofline"] = roofline_block(stages, dom, P, W, H, K, args.kind)
            out["stages_ms"] = {n: round(v["ms"], 4) for n, v in stages.items()}
            out["stages_note"] = ("HIP events inside the library on the launch stream; the roofline kernel (%s) is timed during the "
                                  "timed region, the other stages in an untimed instrumented pre-pass (timing all eight stages "
                                  "costs ~8 %% of the step)" % dom)
            tot_alg = sum(algorithmic_bytes(P, V, R, W, H, K).values())
            out["whole_render"] = {"alg_bytes": tot_alg, "GBps": tot_alg / (ms_per_step * 1e-3) / 1e9,
                                   "frac_of_hbm_peak": tot_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "device_ms_sum": round(sum(v["ms"] for v in stages.values()), 4)}
        # how long the host sat blocked on the device per step (the forward's num_rendered read-back): about one device step =
        # device-bound; near zero while ms_per_step exceeds the device time = this box's host cannot keep the device fed
        out["host"] = {"blocked_on_device_ms_per_step": round(res["host_wait_ms"], 4), "cpu_at_start": int(libc_cpu), "pinned_to_l3_cpus": (len(os.sched_getaffinity(0)) if full_affinity else 0),
                       "note": "ms_per_step - blocked = host-side work per step (Python glue + launches)"}
        if rank_cpus is not None:
            out["host"]["rank_cpus"] = rank_cpus
        # ---- the rest of the north star's workload list, GPU only (no oracle), through the same step ----
        stress = None
        if world == 1 and not args.no_workloads:
            out["workloads"] = []
            import diff_gaussian_rasterization as dgr
            for (w_, h_, p_, k_, kind_, tag) in EXTRA_WORKLOADS:
                ahead = "non-blocking" in tag
                if (w_, h_, p_, k_, kind_) == (W, H, P, K, args.kind) and not ahead and "geometry-only" not in tag:
                    continue
                wl.release()
                wl = Workload(dev, w_, h_, p_, k_, kind_, 0, 1, geo=("geometry-only" in tag))
                dgr.set_async_forward(ahead)
                try:
                    # small workloads take a fraction of a millisecond per step: enough steps for ~50 ms of timed region, and the better
                    # of two timed regions (the first one of a new size also pays the allocator's growth)
                    n_steps = max(20, args.workload_steps) * (1 if p_ >= 1000000 else (3 if p_ >= 300000 else 6))
                    r2 = wl.run(n_steps, 5, profile=not args.no_profile, lib=_C._lib)
                    r3 = wl.run(n_steps, 0, profile=False, lib=_C._lib)
                    if r3["elapsed"] / r3["steps"] < r2["elapsed"] / r2["steps"]:
                        r2["elapsed"], r2["steps"], r2["host_wait_ms"] = r3["elapsed"], r3["steps"], r3["host_wait_ms"]
                finally:
                    dgr.set_async_forward(False)
                e = {"workload": wl.describe(), "tag": tag, "renders_s": r2["steps"] / r2["elapsed"], "ms_per_step": 1e3 * r2["elapsed"] / r2["steps"],
                     "steps": r2["steps"], "timing": "better of two timed regions", "visible": r2["V"], "num_rendered": r2["R"],
                     "host_blocked_on_device_ms_per_step": round(r2["host_wait_ms"], 4)}
                if r2.get("stages"):
                    d_ = r2["dominant"]
                    e.update({"dominant_kernel": d_, "kernel_ms": r2["stages"][d_]["ms"], "frac": r2["stages"][d_]["GBps"] / HBM_PEAK_GBS,
                              "stages_ms": {n: round(v["ms"], 4) for n, v in r2["stages"].items()}})
                out["workloads"].append(e)
                if tag == "stress":
                    stress = (wl, e)
        if world == 1 and not args.no_cpu_baseline:
            if full_affinity:
                os.sched_setaffinity(0, full_affinity)   # the oracle gets every host thread it is allowed
            # the stress configuration's comparison with the oracle at its own size (one oracle render is several seconds of CPU)
            if stress is not None and not args.no_parity and args.cpu_seconds >= 8:
                swl, e = stress
                a2 = argparse.Namespace(cpu_seconds=0.0, width=swl.W, height=swl.H, P=swl.P, K=swl.K)
                _, first = cpu_baseline(a2, swl.sc, swl.cam_cpu, swl.up)
                e["parity"] = parity_block(swl.cam_cpu, swl.sc, swl.up, first, dev, brief=True)
            wl.release()
            wl = Workload(dev, W, H, P, K, args.kind, 0, 1, device_tensors=False, behind_frac=args.behind_frac)
            out["cpu_baseline"], oracle_first = cpu_baseline(args, wl.sc, wl.cam_cpu, wl.up)
            if args.no_parity:
                oracle_first[2].free()
            else:
                out["parity"] = parity_block(wl.cam_cpu, wl.sc, wl.up, oracle_first, dev)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
