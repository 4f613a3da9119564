#!/bin/bash
# which tensors do the Fill / copy kernels of one fused mapping iteration touch?  (grid sizes from the kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ktrace_fill
rm -rf $OUT; mkdir -p $OUT
export HSR_ITER_ONLY=${1:-mapping}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python tools/bench_iteration.py --iters 2 > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
c=collections.Counter()
for r in rows:
    n=r["Kernel_Name"]
    if "Fill" in n or "copyBuffer" in n or "fillBuffer" in n or "direct_copy" in n:
        c[(n[:90], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))]+=1
for k,v in sorted(c.items(), key=lambda kv:-kv[1])[:20]: print(v, k)
PY
find $OUT -name "*.csv" -delete
