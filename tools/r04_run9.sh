cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r04_t9.log 2>&1 || { tail -30 gpurun_out/r04_t9.log; exit 1; }
tail -2 gpurun_out/r04_t9.log
for cfg in "--K 0" "--K 16" "--geo" "--K 26"; do
  for impl in ""; do
      HSR_BWD_IMPL=$impl python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', 'impl=${impl:-q}', 'bwd_render %.4f ms' % d['stages_ms']['bwd_render'], 'step %.3f ms' % d['ms_per_step'])"
  done
done
export HSR_RAST_LIB=$GRAFT_REPO_ROOT/hier-slam_amd/libhsr_rast_trace.so
TRACE_K=0 python tools/trace_bwd.py | grep -v amdgpu | tail -4
