# round 4 end-of-round validation: build() products as shipped, the whole -m gpu suite (default glue, ctypes glue, non-blocking forward), smoke, default bench
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r04_o_suite.full.log 2>&1; grep -v amdgpu.ids gpurun_out/r04_o_suite.full.log | tail -40 > gpurun_out/r04_o_gpu_suite.log; tail -2 gpurun_out/r04_o_gpu_suite.log
HSR_GLUE=ctypes python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r04_o_suite_ctypes.full.log 2>&1; grep -v amdgpu.ids gpurun_out/r04_o_suite_ctypes.full.log | tail -12 > gpurun_out/r04_o_gpu_suite_ctypes_glue.log; tail -2 gpurun_out/r04_o_gpu_suite_ctypes_glue.log
HSR_ASYNC_FORWARD=1 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r04_o_suite_async.full.log 2>&1; grep -v amdgpu.ids gpurun_out/r04_o_suite_async.full.log | tail -12 > gpurun_out/r04_o_gpu_suite_async_forward.log; tail -2 gpurun_out/r04_o_gpu_suite_async_forward.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
python bench.py > gpurun_out/r04_o_bench.json 2> gpurun_out/r04_o_bench.err; python -c "
import json;d=json.load(open('gpurun_out/r04_o_bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['cpu_baseline']['value'],{k:round(v,4) for k,v in d['stages_ms'].items()});print([(w['name'][:28],round(w['value'],1)) for w in d.get('workloads',[])])"
