cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcq; timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d gpurun_out/pmcq -- python bench.py --no-workloads --steps 3 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/pmcq.log 2>&1
python - <<'PY'
import csv, collections, glob
for f in glob.glob("gpurun_out/pmcq/*/*_counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "render_" in k:
            acc["bwd" if "bwd" in k else "fwd"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n in acc: print(n, {c: "%.4g" % (sum(v)/len(v)) for c, v in acc[n].items()})
PY
