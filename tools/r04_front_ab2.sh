cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for m in "fused 1024" "fused 2048" "split 2048" "split 1024"; do
    set -- $m
    for cfg in "" "--P 2000000"; do
      HSR_FRONT=$1 HSR_BIN_CHUNK=$2 python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d['stages_ms'];print('$1 chunk $2', '$cfg', 'front %.4f ms (pre %.4f dup %.4f sort %.4f)' % (s['fwd_preprocess']+s['fwd_duplicate']+s['fwd_sort'], s['fwd_preprocess'], s['fwd_duplicate'], s['fwd_sort']), 'step %.4f ms' % d['ms_per_step'], round(d['value'],1))"
    done
  done
done
