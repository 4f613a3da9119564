# round 4, after the noise-floor model gained the exponent-argument rounding: the whole -m gpu suite, then the 3 000-case run with seed 4242 again
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r04_l_gpu_suite.full.log 2>&1; grep -v amdgpu.ids gpurun_out/r04_l_gpu_suite.full.log | tail -45 > gpurun_out/r04_l_gpu_suite.log; tail -3 gpurun_out/r04_l_gpu_suite.log
HSR_FUZZ_CASES=3000 HSR_FUZZ_SEED=4242 python -m pytest tests/test_gpu_fuzz.py -q -k "test_random_configuration and not legacy and not variants" -p no:cacheprovider > gpurun_out/r04_fuzz4242.full.log 2>&1; grep -v amdgpu.ids gpurun_out/r04_fuzz4242.full.log | tail -40 > gpurun_out/r04_fuzz3000_seed4242.log; tail -3 gpurun_out/r04_fuzz3000_seed4242.log
