#!/bin/bash
# experiment: K <= 27 through the wide kernel template (2 column groups), bf16-split vs fp32 contraction, vs the product kernel
cd $GRAFT_REPO_ROOT
for mode in product subw_f32 subw_bf16; do
  for cfg in "--P 500000 --K 26" "--P 500000 --K 16" "--P 500000 --K 26 --kind aniso"; do
    unset HSR_TEST_BF26
    [ $mode = subw_f32 ] && export HSR_TEST_BF26=0
    [ $mode = subw_bf16 ] && export HSR_TEST_BF26=1
    python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$mode', c['P'], c['K'], c.get('kind'), '%.1f renders/s' % d['value'], 'bwd_render %.3f' % d['stages_ms']['bwd_render'], d.get('parity',{}).get('pass'), d.get('parity',{}).get('grad_max_abs_err_over_max'))"
  done
done
