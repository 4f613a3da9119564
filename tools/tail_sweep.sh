#!/bin/bash
# does the 3.16-round tail (3232 workgroups over 1024 resident slots) show?  Sweeps the image height —
# tiles = 75 x rows — at constant Gaussian density, so the work per tile stays the same and only the number of rounds moves.
# Output: gpurun_out/tail_sweep.jsonl (one bench line per height, with stages_ms.bwd_render / fwd_render).
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/tail_sweep.jsonl
for rows in 27 28 34 35 40 41 42 43 44 46 48 54 55 56; do
  H=$((rows * 16)); P=$((500000 * rows / 43))
  python bench.py --no-workloads --height $H --P $P --steps 60 --warmup 10 --no-cpu-baseline >> gpurun_out/tail_sweep.jsonl 2>/dev/null || echo "rows $rows failed"
done
python - <<'PY'
import json
for l in open("gpurun_out/tail_sweep.jsonl"):
    d = json.loads(l); c = d["config"]; t = ((c["width"] + 15) // 16) * ((c["height"] + 15) // 16)
    s = d["stages_ms"]
    print("tiles %5d  rounds(4 WG/CU) %.3f  R %8d  bwd_render %.4f ms  %.2f ns/tile   fwd_render %.4f ms  %.2f ns/tile   step %.4f" % (
        t, t / 1024.0, c["num_rendered"], s["bwd_render"], 1e6 * s["bwd_render"] / t, s["fwd_render"], 1e6 * s["fwd_render"] / t, d["ms_per_step"]))
PY
