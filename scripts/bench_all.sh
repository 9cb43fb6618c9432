#!/bin/bash
# one bench line per workload (no CPU baseline), summarised
mkdir -p gpurun_out
for w in hubbard_chain_L12_half_filling_U4 hubbard_chain_L14_half_filling_U4 heisenberg_chain_L28_sz0_obc tj_4x5_9up9down_complex hubbard_4x4_7up7down_pbc_U4 hubbard_4x4_half_filling_pbc_U4; do
  LPP_VERBOSE=1 timeout -k 10 500 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/ba_$w.log 2> gpurun_out/ba_$w.err || { tail -5 gpurun_out/ba_$w.err; exit 1; }
  grep "shared-offset" gpurun_out/ba_$w.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ba_$w.log").read().strip().splitlines()[-1])
r=d["roofline"]
print("$w", "it/s %.1f" % d["value"], "spmv_ms %.4f" % r["spmv_ms"], "alg GB/s %.0f" % r["achieved"], "frac %.3f" % r["frac"])
PY
done
