# round 4: the robustness runs EXPERIMENTS §9f asks to repeat after a kernel change: long fuzz runs of both generators, the suite through the
# ctypes glue and with the non-blocking forward forced on
cd $GRAFT_REPO_ROOT
HSR_FUZZ_CASES=3000 HSR_FUZZ_SEED=4242 python -m pytest tests/test_gpu_fuzz.py -q -k "test_random_configuration and not legacy and not variants" -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r04_fuzz3000_seed4242.log; tail -2 gpurun_out/r04_fuzz3000_seed4242.log
HSR_FUZZ_CASES=3000 HSR_FUZZ_SEED=777 python -m pytest tests/test_gpu_fuzz.py -q -k "test_random_configuration and not legacy and not variants" -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r04_fuzz3000_seed777.log; tail -2 gpurun_out/r04_fuzz3000_seed777.log
HSR_FUZZ_CASES_V2=1500 HSR_FUZZ_SEED_V2=31 python -m pytest tests/test_gpu_fuzz.py -q -k "variants" -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r04_fuzz_v2_1500_seed31.log; tail -2 gpurun_out/r04_fuzz_v2_1500_seed31.log
