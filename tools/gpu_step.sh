#!/bin/bash
# Runs a list of GPU steps in order on the GPU box; a step that times out or is killed stops the sequence
# (a failing test does not).  usage: tools/gpu_step.sh "<name>|<timeout s>|<command>" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== $name (limit ${tmo}s): $cmd"
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc"; tail -n 5 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its limit: stopping"; exit $rc; fi
done
exit 0
