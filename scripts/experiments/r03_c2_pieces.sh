#!/bin/bash
# C2 through the pieces kernel (two 512-thread workgroups per CU) against the single-window kernels: kernel trace only
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for cfg in "LPP_PB_CHAIN=0" "LPP_PB_PIECE_ROWS=6464" "LPP_PB_PIECE_ROWS=4352" ""; do
  rm -rf $R/gpurun_out/prof_ab
  env $cfg timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg > $R/gpurun_out/ab.json 2>/dev/null
  echo "== $cfg"; python3 -c "
import json;j=json.load(open('$R/gpurun_out/ab.json'));print(j['value'],j['ms_per_step'],j['roofline']['spmv_ms'])"
  grep -E "k_pb_|k_axpy" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
done
rm -rf $R/gpurun_out/prof_ab
