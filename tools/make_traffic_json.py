#!/usr/bin/env python
"""profiles/<name>.json (tools/summarize_profile.py) -> profiles/traffic_latest.json: the PMC-derived HBM traffic per launch
that bench.py attaches to its roofline object, plus — per tile kernel — what the same counters say bounds it (`limiter`).
usage: python tools/make_traffic_json.py profiles/r02_h_final.json [P W H K kind]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, CLK_GHZ = 1024, 2.0            # 256 CUs x 4 SIMDs; clock held under this load (profiles/r02_valu_rate.jsonl: 1.7-2.2 GHz)
VALU_CYC, MFMA_CYC = 2.17, 32.0       # SIMD cycles per wave64 VALU instruction at 4 waves/SIMD / per v_mfma_f32_16x16x4_f32 (microbenchmark)


def main(src, P=500000, W=1200, H=680, K=26, kind="slam"):
    d = json.load(open(src))
    pmc, ker = d["pmc"], d["kernels"]
    find = lambda sub: next((k for k in pmc if sub in k), None)
    names = {"fwd_render": find("render_fwd_kernel<%d" % K), "bwd_render": find("render_bwd_q_kernel<%d" % K) or find("render_bwd_sub_kernel<%d" % K) or find("render_bwd_subw_kernel"),
             "fwd_preprocess": find("preprocess_kernel"), "bwd_preprocess": find("preprocess_backward_kernel")}
    out = {"workload": {"P": P, "width": W, "height": H, "K": K, "kind": kind},
           "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/profile_gpu.sh)" % os.path.relpath(src, ROOT),
           "formula": "(f*FETCH_SIZE + WRITE_SIZE)*1024 per launch.  f = 2 for the streaming kernels (gfx950 FETCH_SIZE reads half the bytes of a "
                      "16-B-per-lane stream: MI355X_MICROARCH.md, reproduced by tools/micro/fetch_calib.hip: 0.500); f = 1 for the tile kernels, "
                      "whose fetches are per-lane GATHERS of 64-byte records and 4K-byte rows: calibrated on exactly those patterns over tables "
                      "larger than the Infinity Cache, FETCH_SIZE / bytes = 0.98 (records) and 1.08 (rows) — profiles/r03_fetch_calib.json.  "
                      "(Round 2 applied f = 2 to the tile kernels too and over-stated their traffic by the whole fetch side.)",
           "traffic_bytes_per_launch": {}, "raw_fetch_write_KB": {}, "limiter": {}, "bound_by_counters": {}, "issue_frac": {},
           "atomic_floor_ms": {}, "atomic_floor_frac_of_kernel": {}, "hbm_frac_by_traffic": {}}
    for stage, k in names.items():
        if not k or "FETCH_SIZE" not in pmc[k]:
            continue
        c = pmc[k]
        f_fetch = 1.0 if "render" in stage else 2.0
        out["traffic_bytes_per_launch"][stage] = (f_fetch * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        out["raw_fetch_write_KB"][stage] = [c["FETCH_SIZE"], c["WRITE_SIZE"]]
        if "render" in stage and "SQ_INSTS_VALU" in c:
            us = ker[k]["avg_us"]
            cap = SIMDS * us * 1e-6 * CLK_GHZ * 1e9                      # SIMD cycles available during the launch
            valu = (c["SQ_INSTS_VALU"] - c.get("SQ_INSTS_VALU_MFMA_F32", 0.0)) * VALU_CYC
            mfma = c.get("SQ_INSTS_VALU_MFMA_F32", 0.0) * MFMA_CYC
            atom = c.get("TCC_EA0_ATOMIC_sum", 0.0) * 64.0
            txt = ("not HBM-bound (PMC traffic %.0f MB in %.0f us = %.2f TB/s): instruction issue. %.3g VALU + %.3g fp32-MFMA wave-instructions "
                   "= %.0f %% + %.0f %% of the SIMD issue cycles of the launch at the measured rates (%.2f cycles per VALU instruction at 4 waves "
                   "per SIMD — one wave alone issues one per 7 —, %g per MFMA: profiles/r02_valu_rate.jsonl); waves spend %.0f %% of their life "
                   "parked at s_waitcnt / barriers and %.0f %% waiting for an issue slot (SQ_WAIT_ANY, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES); "
                   "LDS bank conflicts %.0f %% of LDS-active cycles (LDS active = %.1f %% of wave cycles)" % (
                       out["traffic_bytes_per_launch"][stage] / 1e6, us, out["traffic_bytes_per_launch"][stage] / us / 1e6,
                       c["SQ_INSTS_VALU"] - c.get("SQ_INSTS_VALU_MFMA_F32", 0.0), c.get("SQ_INSTS_VALU_MFMA_F32", 0.0),
                       100 * valu / cap, 100 * mfma / cap, VALU_CYC, MFMA_CYC, 100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                       100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_ACTIVE_INST_LDS"], 1.0),
                       100 * c["SQ_ACTIVE_INST_LDS"] / c["SQ_WAVE_CYCLES"]))
            if atom:
                txt += ("; %.3g memory-side float-atomic requests x 64 B = %.0f MB, a %.2f ms floor at the guide's 1.3 TB/s chip-wide atomic rate"
                        % (c["TCC_EA0_ATOMIC_sum"], atom / 1e6, atom / 1.3e12 * 1e3))
            out["limiter"][stage] = txt
            issue = (valu + mfma) / cap
            floor_ms = atom / 1.3e12 * 1e3
            hbm = out["traffic_bytes_per_launch"][stage] / (us * 1e-6) / 8e12
            out["issue_frac"][stage] = round(issue, 3)
            out["atomic_floor_ms"][stage] = round(floor_ms, 4)
            out["atomic_floor_frac_of_kernel"][stage] = round(floor_ms / (us * 1e-3), 3)
            out["hbm_frac_by_traffic"][stage] = round(hbm, 3)
            # what the counters say the kernel is closest to: the memory-side atomic pipe, instruction issue, or HBM
            cands = {"atomic pipe": floor_ms / (us * 1e-3), "instruction issue (VALU + matrix)": issue, "hbm": hbm}
            out["bound_by_counters"][stage] = max(cands, key=cands.get) + " (%.2f of the launch; %s)" % (
                max(cands.values()), ", ".join("%s %.2f" % kv for kv in sorted(cands.items(), key=lambda kv: -kv[1])[1:]))
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
    print(json.dumps(out["limiter"], indent=1))


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], *([int(a[1]), int(a[2]), int(a[3]), int(a[4]), a[5]] if len(a) >= 6 else []))
