# timing experiments on the sub-block kernels (HSR_BWD_IMPL=sub, HSR_FWD_IMPL=sub): HSR_DEBUG_FLAGS 1 = no gradient atomics
# (kernel-family / A-B selectors and ablation switches live in the ablate build: make -C hier-slam_amd/csrc ablate)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
export HSR_BWD_IMPL=sub HSR_FWD_IMPL=sub
for f in 0 1; do HSR_DEBUG_FLAGS=$f python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/absub_$f.json && python -c "
import json;d=json.load(open('gpurun_out/absub_$f.json'));print('flags', $f, round(d['value'],1), {a:round(b,4) for a,b in d['stages_ms'].items()})"; done
