# A/B of the matrix-core forward (HSR_FWD_IMPL=mma, hsr_render_fwd_mma.hip) against the per-lane forward: parity cases first, then
# fwd_render at the bench workloads.  usage: bash tools/fwd_mma_ab.sh
# (the matrix-core forward lives in the ablate build: make -C hier-slam_amd/csrc ablate)
export HSR_RAST_LIB=${HSR_RAST_LIB:-$PWD/hier-slam_amd/libhsr_rast_ablate.so} HSR_GLUE=ctypes
set -e
HSR_FWD_IMPL=mma python - <<'PY'
import sys; sys.path[:0]=['hier-slam_amd','tests']
import scenes
from test_gpu_parity import CASES, _compare
for n in ('replica_tree_k26','scannet_tree_k16','generic_k5_white_bg','generic_k40_two_chunks','large_tree_k74','flat_k102','odd_k33','odd_k75_ragged',
          'k52_four_column_groups','k124_widest_single_pass','k130_chunked','wide_deep_tiles_k76','huge_splats','deep_tiles_3000','semantic_k0','culled_behind_camera',
          'config3_scannet_640x480_k16'):
    W,H,P,K,kind,sm,sem,var,bg,beh = CASES[n]
    cam,sc,up = scenes.build(W,H,P,K,seed=11,kind=kind,scale_mult=sm,bg=bg,behind_frac=beh)
    _compare(cam,sc,up,sem,var,None); print('ok', n, flush=True)
PY
run() {  # $1 tag, $2 impl ("" = default), rest: bench flags
    tag=$1; impl=$2; shift 2
    HSR_FWD_IMPL=$impl python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 "$@" > gpurun_out/mma_$tag.json
    python -c "
import json;d=json.load(open('gpurun_out/mma_$tag.json'));print('$tag', round(d['value'],1), 'fwd_render', round(d['stages_ms']['fwd_render'],4))"
}
for cfg in "head" "k16 --K 16" "k74 --K 74" "k102 --K 102" "k48 --K 48" "p100k --P 100000" "p2m --P 2000000 --width 1920 --height 1080" "stress --P 2000000 --width 1920 --height 1080 --K 74" "aniso --kind aniso"; do
    set -- $cfg; tag=$1; shift
    run lane_$tag "" "$@"
    run mma_$tag mma "$@"
done
