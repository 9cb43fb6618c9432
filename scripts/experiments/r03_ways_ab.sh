#!/bin/bash
# A/B of the LDS bank-conflict bound of the template's slot colouring (LPP_PB_BANK_WAYS: lanes of a half-wave allowed on one bank pair)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for w in 2 1 3 4; do
  rm -rf $R/gpurun_out/prof_ab
  LPP_PB_BANK_WAYS=$w timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg > /tmp/ab.json 2>/dev/null
  echo "== ways $w"; grep -E "k_pb_down|k_pb_up" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | sed 's/"void lpp::\(k_pb_[a-z]*\)[^"]*"/\1/' | cut -d, -f1-4
  python3 -c "import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'],d['config']['layout'].get('resident_GB'))"
done
rm -rf $R/gpurun_out/prof_ab
