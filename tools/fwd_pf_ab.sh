#!/bin/bash
# (kernel-family / A-B selectors and ablation switches live in the ablate build: make -C hier-slam_amd/csrc ablate)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "k74 or k75 or k76" 2>&1 | tail -2
for pf in 1 0; do
  for cfg in "--P 500000 --K 74" "--P 2000000 --K 74 --width 1920 --height 1080"; do
    HSR_FWD_PF=$pf python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('pf=$pf', c['P'], c['K'], c['width'], '%.1f renders/s' % d['value'], 'fwd_render', d['stages_ms']['fwd_render'])"
  done
done
