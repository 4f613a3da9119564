# Round 4: the tile-merged rows (MRG) of the Q-panel backward against the per-quadrant rows (HSR_BWD_MERGE=0), same box, alternating.
# HSR_BWD_MERGE: 0 per-quadrant rows, batches of 224 (shipped); 1 merged, batches of 96; 2 merged, batches of 64; 3 per-quadrant rows, batches of 96
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for m in 0 1 2 3; do
    for cfg in "--K 26"; do
      HSR_BWD_MERGE=$m python bench.py --no-workloads --no-cpu-baseline --steps 40 --warmup 8 $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', 'merge=$m', 'bwd_render %.4f ms' % d['stages_ms']['bwd_render'], 'step %.3f ms' % d['ms_per_step'])"
    done
  done
done
