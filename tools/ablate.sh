#!/bin/bash
# usage (on the GPU box): bash tools/ablate.sh "0 2 4 6" [impl]   -> bwd_render / fwd_render ms per HSR_DEBUG_FLAGS value
for f in $1; do
  HSR_DEBUG_FLAGS=$f HSR_BWD_IMPL=${2:-mfma} python bench.py --no-workloads --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/abl.json
  python -c "import json; d=json.load(open('/tmp/abl.json')); print('flags', $f, 'renders/s', round(d['value'],1), d['stages_ms'])"
done
