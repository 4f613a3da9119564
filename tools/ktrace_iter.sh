#!/bin/bash
# per-kernel durations of ONE kind of iteration of tools/bench_iteration.py under rocprofv3 --kernel-trace --stats: the device time of a
# fused mapping (or tracking) iteration next to its host-side wall time.  usage: bash tools/ktrace_iter.sh mapping|tracking
KIND=${1:-mapping}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ktrace_iter_$KIND
rm -rf $OUT; mkdir -p $OUT
export HSR_ITER_ONLY=$KIND
python tools/bench_iteration.py --iters 20 > $OUT/wall.json 2>/dev/null
cat $OUT/wall.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python tools/bench_iteration.py --iters 20 > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/summary.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
# bench_iteration: 3 warm-up + 3 repeats x 20 iterations = 63 iterations of the one kind
n=63.0
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("device time per iteration (sum of kernel durations / 63 iterations): %.3f ms" % (tot/n/1e6))
for r in rows[:30]: print("%-72s calls/iter %5.2f avg %9.1f us  per-iter %7.1f us" % (r["Name"][:72], float(r["Calls"])/n, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/n/1e3))
PY
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" -delete
