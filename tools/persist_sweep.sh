# headline forward stage time against the number of persistent workgroups (HSR_FWD_PERSIST)
export HSR_GLUE=ctypes HSR_FWD_PERSIST_SAY=1
for n in 0 256 512 768 1024 1280 1536 2048 3225; do
    HSR_FWD_PERSIST=$n python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 "$@" > gpurun_out/ps_$n.json 2> gpurun_out/ps_$n.err
    grep -m1 "hsr\]" gpurun_out/ps_$n.err
    python -c "
import json;d=json.load(open('gpurun_out/ps_$n.json'));print($n, round(d['value'],1), round(d['stages_ms']['fwd_render'],4))"
done
for n in 0 768 1024 3225; do
    HSR_FWD_STATIC=1 HSR_FWD_PERSIST=$n python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 "$@" > gpurun_out/pss_$n.json 2> gpurun_out/pss_$n.err
    python -c "
import json;d=json.load(open('gpurun_out/pss_$n.json'));print('static', $n, round(d['value'],1), round(d['stages_ms']['fwd_render'],4))"
done
