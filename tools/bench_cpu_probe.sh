#!/bin/bash
# runs the default bench several times and prints renders/s next to the CPU the main thread sat on (host-side variance hunting)
python - <<'PY'
import os, ctypes
print("allowed cpus:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
os.system("lscpu | grep -E 'NUMA|Model name|^CPU\\(s\\)|Thread' | head -12")
os.system("cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -8 | tr '\\n' ' '; echo")
PY
for i in 1 2 3 4 5 6; do
python - <<'PY'
import subprocess, json, sys, os
r = subprocess.run([sys.executable, "bench.py", "--no-workloads", "--no-cpu-baseline"] + (["--no-pin"] if os.environ.get("NOPIN") else []), capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1])
print(round(d["value"], 1), round(d["ms_per_step"], 4), d["host"])
PY
done
