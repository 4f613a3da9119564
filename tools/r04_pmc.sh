# round 4: instruction-mix and wait counters of the backward tile kernel (usage: bash tools/r04_pmc.sh "<bench flags>" tag)
FLAGS=${1:---K 26}
TAG=${2:-k26}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAVES SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${TAG}_$i
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${TAG}_$i -- python bench.py --no-workloads --steps 3 --warmup 2 --no-cpu-baseline --no-profile $FLAGS > gpurun_out/pmc_${TAG}_$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv, collections, glob, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_${TAG}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "render_bwd" in k:
            acc[k.split("(")[0][-48:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {n: {c: sum(v) / len(v) for c, v in acc[n].items()} for n in acc}
for n in out: print(n, json.dumps({c: float("%.4g" % v) for c, v in sorted(out[n].items())}))
json.dump(out, open("gpurun_out/pmc_${TAG}.json", "w"), indent=1)
PY
