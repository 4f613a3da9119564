# quick A/B: fwd_render of the matrix-core forward vs the per-lane forward at a few widths
# (the matrix-core forward lives in the ablate build: make -C hier-slam_amd/csrc ablate)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
run() { tag=$1; impl=$2; shift 2
    HSR_FWD_IMPL=$impl python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 "$@" > gpurun_out/mma_$tag.json 2>/dev/null
    python -c "
import json;d=json.load(open('gpurun_out/mma_$tag.json'));print('$tag', round(d['value'],1), 'fwd_render', round(d['stages_ms']['fwd_render'],4))"; }
for cfg in "head" "k16 --K 16" "k48 --K 48" "k74 --K 74" "k102 --K 102"; do
    set -- $cfg; tag=$1; shift
    run lane_$tag "" "$@"; run mma_$tag mma "$@"
done
