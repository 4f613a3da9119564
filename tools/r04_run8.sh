cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_truth.py -x -q > gpurun_out/r04_t8.log 2>&1 || { tail -40 gpurun_out/r04_t8.log; exit 1; }
tail -8 gpurun_out/r04_t8.log
export HSR_RAST_LIB=$GRAFT_REPO_ROOT/hier-slam_amd/libhsr_rast_trace.so
python tools/trace_fwd.py 2>&1 | grep -v amdgpu.ids
