# Round 4: A/B of the backward tile kernels on ONE box — round 3's butterfly kernels (HSR_BWD_IMPL=sub) against the Q-panel kernels
# (default).  Alternating, through bench.py's stage timers (HIP events inside the library).  usage: bash tools/r04_bwd_ab.sh [rounds]
set -e
R=${1:-2}
mkdir -p gpurun_out
run() {  # $1 tag, $2 impl ("" = default), rest: bench flags
    tag=$1; impl=$2; shift 2
    HSR_BWD_IMPL=$impl python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 "$@" > gpurun_out/ab_$tag.json
    python -c "
import json,sys;d=json.load(open('gpurun_out/ab_$tag.json'));print('$tag', round(d['value'],1), {a:round(b,4) for a,b in d['stages_ms'].items() if b>0.004})"
}
for r in $(seq 1 $R); do
    run sub_head_$r sub
    run q_head_$r ""
done
for cfg in "k0 --K 0" "k16 --K 16" "geo --geo" "p100k --P 100000" "p2m --P 2000000" "aniso --kind aniso"; do
    set -- $cfg; tag=$1; shift
    run sub_$tag sub "$@"
    run q_$tag "" "$@"
done
