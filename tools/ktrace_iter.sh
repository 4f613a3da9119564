#!/bin/bash
# per-kernel durations of tools/bench_iteration.py (mapping + tracking iterations) under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ktrace_iter
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python tools/bench_iteration.py --iters 10 > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:26]: print("%-64s calls %4s avg %9.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3))
PY
