#!/bin/bash
# A/B of whole-line (16 positions) against half-line (8 positions) coupling panels at the (7,6) sector (kernel trace only)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for cfg in ${CFGS:-"LPP_PB_HALF=0" "LPP_PB_HALF=1"}; do
  rm -rf $R/gpurun_out/prof_ab
  env $cfg timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --engine onthefly --workload ${WL:-hubbard_4x5_7up6down_pbc_U4} --steps 4 --warmup 1 --no-cpu-baseline --no-reortho-leg > /tmp/ab.json 2>/dev/null
  echo "== $cfg"; grep -E "k_pb_down|k_pb_up_big|k_pb_combine" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | sed 's/"[a-z ]*lpp::\(k_pb_[a-z0-9_]*\)\([^"]*\)"/\1\2/' | awk -F, '{print $1,$(NF-5),$(NF-4),$(NF-3)}' 
  python3 -c "import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'],'e0',d.get('e0_after_steps'))"
done
rm -rf $R/gpurun_out/prof_ab
