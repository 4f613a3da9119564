# round 4: the opt-in exact semantic -> alpha mode through the parity cases and 600 fuzz cases (library and oracle both in that mode)
cd $GRAFT_REPO_ROOT
export HSR_TEST_SEM_ALPHA=exact
python -m pytest tests/test_gpu_parity.py -q -k "test_parity" -p no:cacheprovider > gpurun_out/r04_s_exact_parity.full.log 2>&1; grep -v amdgpu.ids gpurun_out/r04_s_exact_parity.full.log | tail -6
HSR_FUZZ_CASES=600 HSR_FUZZ_SEED=99 python -m pytest tests/test_gpu_fuzz.py -q -k "test_random_configuration and not legacy and not variants" -p no:cacheprovider > gpurun_out/r04_s_exact_fuzz.full.log 2>&1; grep -v amdgpu.ids gpurun_out/r04_s_exact_fuzz.full.log | tail -30 > gpurun_out/r04_s_exact_mode_fuzz600_seed99.log; tail -4 gpurun_out/r04_s_exact_mode_fuzz600_seed99.log
HSR_FUZZ_CASES_V2=300 HSR_FUZZ_SEED_V2=7 python -m pytest tests/test_gpu_fuzz.py -q -k "variants" -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3
