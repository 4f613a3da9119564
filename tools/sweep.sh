#!/bin/bash
# SURVEY.md §8(d) workload matrix: one bench.py line per configuration into gpurun_out/sweep_<tag>.jsonl
set -e
tag=${1:-r01}
out=gpurun_out/sweep_${tag}.jsonl
mkdir -p gpurun_out; : > $out
run() { python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 "$@" >> $out; echo "done $*"; }
run --P 100000 --K 26
run --P 300000 --K 26
run --P 500000 --K 26
run --P 2000000 --K 26
run --P 500000 --K 16
run --P 500000 --K 26 --kind aniso
run --P 500000 --K 0
run --P 500000 --K 74
run --P 500000 --K 102
run --P 2000000 --K 74 --width 1920 --height 1080
