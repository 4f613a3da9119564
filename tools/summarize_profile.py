#!/usr/bin/env python
"""Condenses a tools/profile_gpu.sh output directory into one JSON + one text table under profiles/.
usage: python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name>"""
import collections
import csv
import glob
import json
import os
import sys


def short(k):
    for a, b in (("(anonymous namespace)::", ""), ("void ", "")):
        k = k.replace(a, b)
    return k.split("(")[0][:60]


def main(src, dst):
    out = {"kernels": {}, "pmc": {}}
    for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            out["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                               "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            for c, v in d.items():
                out["pmc"].setdefault(k, {})[c] = sum(v) / len(v)  # average per launch
    # HBM traffic per launch: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads half of a wide coalesced
    # stream (MI355X_MICROARCH.md §HBM) -> doubled; gathers / sub-16B accesses are uncalibrated (stated with the number).
    for k, d in out["pmc"].items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_traffic_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
            d["hbm_traffic_note"] = "(2*FETCH_SIZE + WRITE_SIZE)*1024; gfx950 FETCH_SIZE x2 correction for 16B/lane streams; gather patterns uncalibrated"
    json.dump(out, open(dst + ".json", "w"), indent=1, sort_keys=True)
    with open(dst + ".txt", "w") as fh:
        fh.write("%-62s %7s %10s %10s %6s\n" % ("kernel", "calls", "avg_us", "total_ms", "pct"))
        for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
            fh.write("%-62s %7d %10.2f %10.3f %6.2f\n" % (k, v["calls"], v["avg_us"], v["total_ms"], v["pct"]))
        fh.write("\nPMC (average per launch)\n")
        for k, d in sorted(out["pmc"].items()):
            if "render" in k or "sort" in k or "preprocess" in k or "duplicate" in k:
                fh.write(k + "\n")
                for c, v in sorted(d.items()):
                    if isinstance(v, float):
                        fh.write("    %-36s %.6g\n" % (c, v))
    print(open(dst + ".txt").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
