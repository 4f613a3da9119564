# round 4: A/B of the XCD -> tile mapping (strip per XCD, the product, vs a 2-D region per XCD: make -C hier-slam_amd/csrc map2d), same box,
# alternating, through the ctypes glue.  First the parity cases through the experiment library (a wrong mapping renders wrong tiles).
cd $GRAFT_REPO_ROOT
export HSR_GLUE=ctypes
HSR_RAST_LIB=$PWD/hier-slam_amd/libhsr_rast_map2d.so python -m pytest tests/test_gpu_parity.py -x -q -k "test_parity" 2>&1 | tail -1
run() {
    tag=$1; lib=$2; shift 2
    HSR_RAST_LIB=$PWD/hier-slam_amd/$lib python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$tag', round(d['value'],1), {a:round(b,4) for a,b in d['stages_ms'].items() if 'render' in a})"
}
for r in 1 2 3; do
    run strip_$r libhsr_rast.so
    run map2d_$r libhsr_rast_map2d.so
done
for cfg in "k74 --K 74" "p2m --P 2000000" "aniso --kind aniso"; do
    set -- $cfg; tag=$1; shift
    run strip_$tag libhsr_rast.so "$@"
    run map2d_$tag libhsr_rast_map2d.so "$@"
done
