set -e
for k in 74 102; do python bench.py --no-workloads --no-cpu-baseline --steps 30 --warmup 5 --P 500000 --K $k > gpurun_out/k$k.json; done
python - <<'PY'
import json
for k in (74,102):
    d=json.load(open("gpurun_out/k%d.json"%k)); print(k, round(d["value"],1), {a:round(b,3) for a,b in d["stages_ms"].items()})
PY
