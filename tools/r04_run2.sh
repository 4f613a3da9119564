# round 4, experiment 2: the backward tile kernels with and without their gradient atomics (diagnostic build), old (sub) vs new (q),
# and the phase timers of the new kernel
cd $GRAFT_REPO_ROOT
export HSR_RAST_LIB=$GRAFT_REPO_ROOT/hier-slam_amd/libhsr_rast_ablate.so HSR_GLUE=ctypes
for cfg in "--K 26" "--K 0" "--geo"; do
  for impl in sub ""; do
    for f in 0 1; do
      HSR_BWD_IMPL=$impl HSR_DEBUG_FLAGS=$f python bench.py --no-workloads --no-cpu-baseline --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', 'impl=${impl:-q}', 'atomics dropped' if $f else 'atomics on     ', 'bwd_render %.4f ms' % d['stages_ms']['bwd_render'], 'step %.3f ms' % d['ms_per_step'])"
    done
  done
done
export HSR_RAST_LIB=$GRAFT_REPO_ROOT/hier-slam_amd/libhsr_rast_trace.so
python tools/trace_bwd.py
TRACE_GEO=1 python tools/trace_bwd.py
TRACE_K=0 python tools/trace_bwd.py
HSR_BWD_IMPL=sub python tools/trace_bwd.py
