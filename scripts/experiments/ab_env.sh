#!/bin/bash
# A/B an environment switch over a few workloads: ab_env.sh VAR "w1 w2 ..."
VAR=$1; shift
mkdir -p gpurun_out
for w in $@; do
  for v in 1 0; do
    env $VAR=$v timeout -k 10 500 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/ab.log 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
print("$w", "$VAR=$v", "it/s %.1f" % d["value"], "spmv_ms %.4f" % d["roofline"]["spmv_ms"])
PY
  done
done
