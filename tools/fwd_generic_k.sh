#!/bin/bash
# forward at "any other K": per-lane sub-block kernel with L2-touch staging (default for 27..80) vs the matrix-core kernel (HSR_FWD_IMPL=wide)
# (since round 3 HSR_BWD_IMPL=mfma / HSR_FWD_IMPL=wide exist in the ablate build only: `make -C hier-slam_amd/csrc ablate`)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "k33 or k40 or k74 or k75 or k76 or k5 or k124 or k130" 2>&1 | tail -2
for impl in default wide; do
  for K in 27 28 32 33 43 48 49 60 64 65 75 80 81; do
    if [ $impl = wide ]; then export HSR_FWD_IMPL=wide; else unset HSR_FWD_IMPL; fi
    python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 --K $K 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$impl K=$K', round(d['value'],1), 'fwd_render', d['stages_ms']['fwd_render'])"
  done
done
