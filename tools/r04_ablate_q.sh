# round 4: where the Q-panel backward's time goes — timing ablations of the diagnostic build (results are wrong with a flag set; timing only)
# flags (HSR_DEBUG_FLAGS): 1 no gradient atomics, 2 no panel stores, 4 no median add, 8 no flush, 32 no moments / emission, 64 no matrix instructions,
# 256 no visit loop, 512 no record reads in the loop, 1024 no v_exp / v_rcp, 2048 no list-element reads
cd $GRAFT_REPO_ROOT
export HSR_RAST_LIB=$GRAFT_REPO_ROOT/hier-slam_amd/libhsr_rast_ablate.so HSR_GLUE=ctypes
for cfg in ${CFGS:-"--geo" "--K 0"}; do
    for f in ${FLAGS:-0 1 3 5 9 257 265 513 1025 2049 2563 3587 3591}; do
      HSR_DEBUG_FLAGS=$f python bench.py --no-workloads --no-cpu-baseline --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', 'flags=%4d' % $f, 'bwd_render %.4f ms' % d['stages_ms']['bwd_render'])"
    done
done
