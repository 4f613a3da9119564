# forward-stage time, wide matrix-core kernel vs per-lane kernels, for several K
# (kernel-family / A-B selectors and ablation switches live in the ablate build: make -C hier-slam_amd/csrc ablate)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
for k in 33 40 60 74 90 102 124; do
  for impl in wide valu; do
    if [ $impl = valu ]; then export HSR_FWD_IMPL=valu; else unset HSR_FWD_IMPL; fi
    python bench.py --no-workloads --no-cpu-baseline --steps 15 --warmup 3 --P 500000 --K $k > gpurun_out/kc.json
    python -c "import json;d=json.load(open('gpurun_out/kc.json'));print('K=$k $impl fwd_render %.3f ms  bwd_render %.3f' % (d['stages_ms']['fwd_render'], d['stages_ms']['bwd_render']))"
  done
done
