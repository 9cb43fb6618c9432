#!/bin/bash
mkdir -p gpurun_out
w=$1; shift
for r in $@; do
  LPP_SPMV_KERNEL=3 LPP_WINDOW_ROWS=$r timeout -k 10 500 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/ab.log 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
print("$w", "rows=$r", "it/s %.1f" % d["value"], "spmv_ms %.4f" % d["roofline"]["spmv_ms"], d["config"]["layout"]["per_row_entries"], d["config"]["layout"]["shared_offset_entries"])
PY
done
