#!/bin/bash
# product-basis layout vs general layout on the smaller Hubbard workloads (where does the product layout start to pay?)
mkdir -p gpurun_out
for w in hubbard_chain_L12_half_filling_U4 hubbard_chain_L14_half_filling_U4 hubbard_4x4_7up7down_pbc_U4; do
  for v in 1 0; do
    LPP_PRODUCT_LAYOUT=$v timeout -k 10 300 python bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline --no-generic-csr --no-e0-check > gpurun_out/pbt.log 2> gpurun_out/pbt.err || { tail -5 gpurun_out/pbt.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/pbt.log").read().strip().splitlines()[-1])
print("$w", "product=$v", "rows", d["config"]["rows"], "it/s %.1f" % d["value"], "ms/step %.4f" % d["ms_per_step"], "spmv_ms %.4f" % d["roofline"]["spmv_ms"], d["config"]["layout"]["kernel"])
PY
  done
done
