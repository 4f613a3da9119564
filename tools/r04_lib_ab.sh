# A/B of two builds of the library on one box (libhsr_rast_prev.so = the previous build, copied by hand), alternating, ctypes glue
cd $GRAFT_REPO_ROOT
export HSR_GLUE=ctypes
run() {
    tag=$1; lib=$2; shift 2
    HSR_RAST_LIB=$PWD/hier-slam_amd/$lib python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$tag', round(d['value'],1), {a:round(b,4) for a,b in d['stages_ms'].items() if 'bwd_render' in a or 'fwd_render' in a})"
}
for cfg in "k26" "k0 --K 0" "k16 --K 16" "geo --geo" "aniso --kind aniso" "p2m --P 2000000"; do
    set -- $cfg; tag=$1; shift
    for r in 1 2; do
    run prev_$tag libhsr_rast_prev.so "$@"
    run new_$tag libhsr_rast.so "$@"
    done
done
