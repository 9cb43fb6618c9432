#!/bin/bash
# A/B of the coupling kernel over parts at the (7,6) sector: pacing on/off, number of parts (kernel trace only)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for cfg in "" "LPP_PB_PACE=0" "LPP_PB_PARTS=1" "LPP_PB_PARTS=2" "LPP_PB_PARTS=4" "LPP_PB_PARTS=6"; do
  rm -rf $R/gpurun_out/prof_ab
  env $cfg timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --engine onthefly --workload ${WL:-hubbard_4x5_7up6down_pbc_U4} --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== $cfg"; grep -E "k_pb_down_parts|k_pb_up_big" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
rm -rf $R/gpurun_out/prof_ab
