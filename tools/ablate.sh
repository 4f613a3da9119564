#!/bin/bash
# usage (on the GPU box): bash tools/ablate.sh "0 2 4 6" [impl]   -> bwd_render / fwd_render ms per HSR_DEBUG_FLAGS value
# (kernel-family / A-B selectors and ablation switches live in the ablate build: make -C hier-slam_amd/csrc ablate)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
for f in $1; do
  HSR_DEBUG_FLAGS=$f HSR_BWD_IMPL=${2:-mfma} python bench.py --no-workloads --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/abl.json
  python -c "import json; d=json.load(open('/tmp/abl.json')); print('flags', $f, 'renders/s', round(d['value'],1), d['stages_ms'])"
done
