# A/B of two builds of the library on the same box: hier-slam_amd/libhsr_rast_prev.so (the previous revision, built by hand from
# `git archive`) against the tree's libhsr_rast.so, alternating, through the ctypes glue (HSR_RAST_LIB selects the library).
# usage: bash tools/lib_ab.sh [rounds]
set -e
R=${1:-2}
export HSR_GLUE=ctypes
run() {  # $1 tag, $2 lib, rest: bench flags
    tag=$1; lib=$2; shift 2
    HSR_RAST_LIB=$PWD/hier-slam_amd/$lib python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 "$@" > gpurun_out/ab_$tag.json
    python -c "
import json,sys;d=json.load(open('gpurun_out/ab_$tag.json'));print('$tag', round(d['value'],1), {a:round(b,4) for a,b in d['stages_ms'].items() if b>0.004})"
}
for r in $(seq 1 $R); do
    run prev_head_$r libhsr_rast_prev.so
    run new_head_$r libhsr_rast.so
done
for cfg in "k74 --K 74" "k102 --K 102" "k16 --K 16" "p2m --P 2000000 --width 1920 --height 1080" "p100k --P 100000" "stress --P 2000000 --width 1920 --height 1080 --K 74" "aniso --kind aniso"; do
    set -- $cfg; tag=$1; shift
    run prev_$tag libhsr_rast_prev.so "$@"
    run new_$tag libhsr_rast.so "$@"
done
