#!/bin/bash
# A/B of the tile-kernel families over the workload matrix: default (4x4 sub-block lists) vs HSR_FWD_IMPL=valu HSR_BWD_IMPL=mfma
# (quadrant lists).  Prints renders/s and the two render stages per configuration.
# (since round 3 HSR_BWD_IMPL=mfma / HSR_FWD_IMPL=wide exist in the ablate build only: `make -C hier-slam_amd/csrc ablate`)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
run() { python bench.py --no-workloads --no-cpu-baseline --steps 20 --warmup 5 "$@" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %-44s %8.1f  fwd %.3f bwd %.3f' % ('$TAGX', '$*', d['value'], d['stages_ms']['fwd_render'], d['stages_ms']['bwd_render']))"; }
for cfg in "--P 100000 --K 26" "--P 2000000 --K 26" "--P 500000 --K 16" "--P 500000 --K 26 --kind aniso" "--P 500000 --K 0" "--P 500000 --K 20"; do
  TAGX=sub; run $cfg
  export HSR_FWD_IMPL=valu HSR_BWD_IMPL=mfma; TAGX=quad; run $cfg; unset HSR_FWD_IMPL HSR_BWD_IMPL
done
