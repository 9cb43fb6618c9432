#!/bin/bash
mkdir -p gpurun_out
for w in $@; do
  for k in 0 2 3; do
    LPP_SPMV_KERNEL=$k timeout -k 10 500 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/ab.log 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
print("$w", "kernel=$k", "it/s %.1f" % d["value"], "spmv_ms %.4f" % d["roofline"]["spmv_ms"], d["config"]["layout"]["kernel"], d["config"]["layout"]["block_template"])
PY
  done
done
