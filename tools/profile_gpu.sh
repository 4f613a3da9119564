#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes over bench.py.
# Output: gpurun_out/prof_<tag>/...  Post-process with tools/summarize_profile.py, copy results to profiles/.
# Counters go in their own runs (no trace domains mixed in), FETCH_SIZE and WRITE_SIZE in separate passes
# (TCC slot limits, MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -e
TAG=${1:-run}
ARGS=${2:-"--steps 10 --warmup 3 --no-cpu-baseline --no-profile"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --no-workloads $ARGS > $OUT/trace.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_F32" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_IFETCH"; do
  n=$(echo "$set" | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$n -- python bench.py --no-workloads $ARGS > $OUT/pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
echo profiled $TAG
