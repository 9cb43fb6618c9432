#!/bin/bash
# A/B of the block order inside a coupling workgroup's range: sorted by list length over the whole range (default) or inside
# windows of consecutive blocks (LPP_PB_ORDER_WINDOW) -- consecutive blocks share source lines (L1 hits).  Kernel trace only.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for w in 0 32 48 64 80 96 0 64; do
  rm -rf $R/gpurun_out/prof_ab
  env LPP_PB_ORDER_WINDOW=$w timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg > /tmp/ab.json 2>/dev/null
  echo "== window $w"; grep -E "k_pb_down|k_pb_up" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | sed 's/"[^"]*"/K/' | cut -d, -f1-4
  python3 -c "import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'])"
done
rm -rf $R/gpurun_out/prof_ab
