#!/bin/bash
# What the backward tile kernel costs with and without its gradient atomics (diagnostic build: HSR_DEBUG_FLAGS=1 drops them), per
# workload: how far the kernel sits above / below its atomic floor, and what a smaller number of rows could buy at most.
# usage (GPU box, after `make -C hier-slam_amd/csrc ablate`): bash tools/atomics_ablate.sh > gpurun_out/atomics_ablate.log
export HSR_RAST_LIB=$GRAFT_REPO_ROOT/hier-slam_amd/libhsr_rast_ablate.so HSR_GLUE=ctypes
for cfg in "--P 500000 --K 26" "--P 500000 --K 74" "--P 500000 --K 102" "--P 2000000 --K 74 --width 1920 --height 1080"; do
  for f in 0 1; do
    HSR_DEBUG_FLAGS=$f python bench.py --no-workloads --no-cpu-baseline --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', 'atomics dropped' if $f else 'atomics on     ', 'bwd_render %.4f ms' % d['stages_ms']['bwd_render'], 'step %.3f ms' % d['ms_per_step'])"
  done
done
