#!/bin/bash
# A/B of the look-ahead split between the two value groups in k_pb_up<CHAIN> (LPP_PB_PRE0 = chunks of group 0; group 1 gets 8 - PRE0)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for w in auto auto; do
  rm -rf $R/gpurun_out/prof_ab
  if [ $w = auto ]; then unset LPP_PB_PRE0; else export LPP_PB_PRE0=$w; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg > /tmp/ab.json 2>/dev/null
  echo "== PRE0 $w"; grep -E "k_pb_down|k_pb_up" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | sed 's/"void lpp::\(k_pb_[a-z]*\)[^"]*"/\1/' | cut -d, -f1-4
  python3 -c "import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'],d['config']['coefficients_vs_cpu_oracle']['max_rel_diff'])"
done
rm -rf $R/gpurun_out/prof_ab
