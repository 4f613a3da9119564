# Round 4: the fused front end (preprocess + tile count in one launch; default) against the three launches (HSR_FRONT=split), same box, alternating
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden_and_scale.py -x -q -p no:cacheprovider > gpurun_out/r04_front_tests.log 2>&1 || { grep -v amdgpu.ids gpurun_out/r04_front_tests.log | tail -30; exit 1; }
tail -1 gpurun_out/r04_front_tests.log
for r in 1 2 3; do
  for m in fused split; do
    for cfg in "" "--P 100000" "--P 2000000"; do
      HSR_FRONT=$m python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d['stages_ms'];print('$m', '$cfg', 'front %.4f ms (pre %.4f dup %.4f sort %.4f)' % (s['fwd_preprocess']+s['fwd_duplicate']+s['fwd_sort'], s['fwd_preprocess'], s['fwd_duplicate'], s['fwd_sort']), 'step %.4f ms' % d['ms_per_step'], round(d['value'],1))"
    done
  done
done
