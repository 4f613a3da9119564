# PMC counters of the render kernels for a given K (usage: bash tools/pmc_k.sh 102)
K=${1:-102}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE"; do
  n=$(echo "$set" | cut -d' ' -f1)
  rm -rf gpurun_out/pmck_$n
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmck_$n -- python bench.py --no-workloads --steps 3 --warmup 2 --no-cpu-baseline --no-profile --K $K > gpurun_out/pmck_$n.log 2>&1 || echo "pass $n failed"
done
python - <<'PY'
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmck_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "render_" in k:
            acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in acc: print(n, {c: "%.4g" % (sum(v)/len(v)) for c, v in acc[n].items()})
PY
