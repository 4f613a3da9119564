# forward-stage time of the wide kernel under ablation flags (HSR_FWD_DEBUG): usage bash tools/ablate_wide.sh 102
K=${1:-102}
for f in 0 1 2 3 4 6; do
  HSR_FWD_DEBUG=$f python bench.py --no-workloads --no-cpu-baseline --steps 20 --warmup 3 --P 500000 --K $K > gpurun_out/abl_$f.json
  python -c "import json;d=json.load(open('gpurun_out/abl_$f.json'));print('flags $f fwd_render %.3f ms' % d['stages_ms']['fwd_render'])"
done
