#!/bin/bash
# experiment: the per-lane (PF, quad-shared rows) forward beyond 80 channels vs the matrix-core forward (hsr_render_fwd_wide.hip)
# (since round 3 HSR_BWD_IMPL=mfma / HSR_FWD_IMPL=wide exist in the ablate build only: `make -C hier-slam_amd/csrc ablate`)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
cd $GRAFT_REPO_ROOT
for mode in wide_mfma per_lane; do
  for K in 90 102 124; do  # (the per_lane rows need HSR_FWD_PF_MAX, an experiment switch that was removed when the result became the default)
    if [ $mode = per_lane ]; then export HSR_FWD_PF_MAX=128; else unset HSR_FWD_PF_MAX; fi
    python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 --P 500000 --K $K 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$mode', c['K'], '%.1f renders/s' % d['value'], 'fwd_render %.3f' % d['stages_ms']['fwd_render'], d.get('parity',{}).get('pass') if d.get('parity') else None)"
  done
done
