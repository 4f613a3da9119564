# round 4: parity of the Q-panel backward (median table, compact rows) + timing against round 3's kernels
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_truth.py tests/test_gpu_golden_and_scale.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r04_t6.log 2>&1 || { tail -40 gpurun_out/r04_t6.log; exit 1; }
tail -4 gpurun_out/r04_t6.log
for cfg in "--K 26" "--K 16" "--K 0" "--geo"; do
  for impl in sub ""; do
      HSR_BWD_IMPL=$impl python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', 'impl=${impl:-q}', 'bwd_render %.4f ms' % d['stages_ms']['bwd_render'], 'bwd_zero %.4f' % d['stages_ms']['bwd_zero'], 'bwd_pre %.4f' % d['stages_ms']['bwd_preprocess'], 'step %.3f ms' % d['ms_per_step'])"
  done
done
export HSR_RAST_LIB=$GRAFT_REPO_ROOT/hier-slam_amd/libhsr_rast_trace.so
TRACE_GEO=1 python tools/trace_bwd.py
TRACE_K=0 python tools/trace_bwd.py
