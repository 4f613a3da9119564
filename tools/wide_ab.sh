#!/bin/bash
# A/B of the wide-tree backward: one pass (default) vs 64-column passes (HSR_BWD_WIDE_PASS=split); parity first.
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "k74 or k102 or k75 or k76 or k124 or k130 or k40 or k33" 2>&1 | tail -3
for mode in single split; do
  for cfg in "--P 500000 --K 74" "--P 500000 --K 102" "--P 2000000 --K 74 --width 1920 --height 1080"; do
    if [ $mode = split ]; then export HSR_BWD_WIDE_PASS=split; else unset HSR_BWD_WIDE_PASS; fi
    python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('$mode', c['P'], c['K'], c['width'], '%.1f renders/s' % d['value'], d['stages_ms'])"
  done
done
