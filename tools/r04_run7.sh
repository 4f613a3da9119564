# round 4: new evidence tests on the GPU + the 2-rank rehearsal of the multi-GPU step (gloo, one shared GPU)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_map_io.py tests/test_gpu_fuzz.py tests/test_gpu_slam_loop.py -x -q > gpurun_out/r04_t7.log 2>&1 || { tail -40 gpurun_out/r04_t7.log; exit 1; }
tail -12 gpurun_out/r04_t7.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
HSR_BENCH_SHARE_GPU=1 HSR_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-workloads > gpurun_out/r04_gloo2.json 2> gpurun_out/r04_gloo2.err || { tail -20 gpurun_out/r04_gloo2.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r04_gloo2.json'));print({k:d[k] for k in ('value','n_gpus','ms_per_step')}, d['config']['exchange_mode'], d['config']['keyframes_per_rank'], {k:d['exchange'][k] for k in ('steps','keyframes_per_step','zero_copy_tensors','copied_tensors','union_fraction','exchange_bytes_per_step')})"
HSR_BENCH_SHARE_GPU=1 HSR_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-workloads --stale-gradients --keyframes-per-rank 1 > gpurun_out/r04_gloo2_stale.json 2> gpurun_out/r04_gloo2_stale.err || { tail -20 gpurun_out/r04_gloo2_stale.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r04_gloo2_stale.json'));print({k:d[k] for k in ('value','n_gpus','ms_per_step')}, d['config']['exchange_mode'])"
