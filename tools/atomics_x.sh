# timing experiments on the backward (not a product path): the quadrant-list matrix-core kernel (HSR_BWD_IMPL=mfma) with
# 0x / 1x / 2x / 3x the atomic row requests (HSR_DEBUG_FLAGS 1 / 0 / 8 / 24), and the per-instance-rows kernel (no global atomics)
# (since round 3 HSR_BWD_IMPL=mfma / HSR_FWD_IMPL=wide exist in the ablate build only: `make -C hier-slam_amd/csrc ablate`)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
for f in 1 0 8 24; do HSR_BWD_IMPL=mfma HSR_DEBUG_FLAGS=$f python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/atomx_$f.json && python -c "
import json;d=json.load(open('gpurun_out/atomx_$f.json'));print('flags', $f, round(d['value'],1), {a:round(b,4) for a,b in d['stages_ms'].items()})"; done
HSR_BWD_IMPL=rows python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/atomx_rows.json && python -c "
import json;d=json.load(open('gpurun_out/atomx_rows.json'));print('rows', round(d['value'],1), {a:round(b,4) for a,b in d['stages_ms'].items()})"
