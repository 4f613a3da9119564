#!/bin/bash
# per-kernel average durations of the default bench (rocprofv3 --kernel-trace --stats), top 16 rows -> stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ktrace_$1
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --no-workloads --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:16]: print("%-70s calls %4s avg %9.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
