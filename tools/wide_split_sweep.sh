#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.readline()); c=d['config']
print(os.environ.get('HSR_BWD_WIDE_FIRST','-'), os.environ.get('HSR_BWD_WIDE_CHUNK','-'), c['P'], c['K'], c['width'], '%.1f renders/s' % d['value'], 'bwd_render', d['stages_ms']['bwd_render'])"; }
for K in 74 102; do
  unset HSR_BWD_WIDE_FIRST HSR_BWD_WIDE_CHUNK; run --P 500000 --K $K
  for f in 11 27 43; do for ch in 32 48 64 107; do HSR_BWD_WIDE_FIRST=$f HSR_BWD_WIDE_CHUNK=$ch run --P 500000 --K $K; done; done
done
