#!/bin/bash
# A/B of the wide-tree backward's panel contraction: bf16 matrix cores on the exact three-way split (default, 33 <= K + 5 <= 80)
# vs fp32 matrix instructions (HSR_BWD_WIDE_MMA=f32); parity first.
# (kernel-family / A-B selectors and ablation switches live in the ablate build: make -C hier-slam_amd/csrc ablate)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "k74 or k102 or k75 or k76 or k124 or k130 or k40 or k33" 2>&1 | tail -3
for mode in bf16x3 f32; do
  for cfg in "--P 500000 --K 32" "--P 500000 --K 43" "--P 500000 --K 59" "--P 500000 --K 74" "--P 2000000 --K 74 --width 1920 --height 1080"; do
    if [ $mode = f32 ]; then export HSR_BWD_WIDE_MMA=f32; else unset HSR_BWD_WIDE_MMA; fi
    python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$mode', c['P'], c['K'], c['width'], '%.1f renders/s' % d['value'], 'bwd_render %.3f' % d['stages_ms']['bwd_render'], d.get('parity',{}).get('pass'))"
  done
done
