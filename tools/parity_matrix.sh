#!/bin/bash
# bench.py WITH its oracle leg (cpu_baseline + parity at the configuration's own size) over the SURVEY §8(d) matrix -> gpurun_out/parity_matrix.jsonl
cd $GRAFT_REPO_ROOT
out=gpurun_out/parity_matrix.jsonl; : > $out
run() { python bench.py --no-workloads --steps 20 --warmup 5 "$@" >> $out 2>/dev/null; echo "done $*"; }
run --P 500000 --K 26
run --P 500000 --K 26 --kind aniso
run --P 500000 --K 16
run --P 500000 --K 0
run --P 500000 --K 74
run --P 500000 --K 102
run --P 2000000 --K 26
python - <<'PY'
import json
for l in open("gpurun_out/parity_matrix.jsonl"):
    d = json.loads(l); c = d["config"]; p = d["parity"]
    print(c["P"], c["K"], c.get("kind", "slam"), "%.0f renders/s" % d["value"], "pass", p["pass"], "with tie bounds", p["pass_with_tie_bounds"],
          "tie-risk pixels", p["oracle_tie_risk_pixels"], "max grad err/max %.1e" % max(p["grad_err_over_max"].values()),
          "beyond bound %.1e" % max(p["grad_err_over_max_beyond_tie_bound"].values()), "ints", all(p[k] for k in p if k.endswith("_equal")))
PY
