set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_truth.py tests/test_gpu_golden_and_scale.py -x -q > gpurun_out/r04_t1.log 2>&1 || { tail -40 gpurun_out/r04_t1.log; exit 1; }
tail -15 gpurun_out/r04_t1.log
bash tools/r04_bwd_ab.sh 2 2>&1 | tee gpurun_out/r04_ab1.log
