# round 4, experiment 3: parity of the pipelined Q-panel backward, its timing against round 3's kernel, atomics on / off, phase timers
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r04_t3.log 2>&1 || { tail -40 gpurun_out/r04_t3.log; exit 1; }
tail -3 gpurun_out/r04_t3.log
for cfg in "--K 26" "--K 0" "--geo"; do
  for impl in sub ""; do
      HSR_BWD_IMPL=$impl python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', 'impl=${impl:-q}', 'bwd_render %.4f ms' % d['stages_ms']['bwd_render'], 'step %.3f ms' % d['ms_per_step'])"
  done
done
export HSR_RAST_LIB=$GRAFT_REPO_ROOT/hier-slam_amd/libhsr_rast_ablate.so HSR_GLUE=ctypes
for cfg in "--K 26" "--K 0" "--geo"; do
    for f in 0 1; do
      HSR_DEBUG_FLAGS=$f python bench.py --no-workloads --no-cpu-baseline --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', 'impl=q (ablate build)', 'atomics dropped' if $f else 'atomics on     ', 'bwd_render %.4f ms' % d['stages_ms']['bwd_render'], 'step %.3f ms' % d['ms_per_step'])"
    done
done
export HSR_RAST_LIB=$GRAFT_REPO_ROOT/hier-slam_amd/libhsr_rast_trace.so
python tools/trace_bwd.py
TRACE_GEO=1 python tools/trace_bwd.py
