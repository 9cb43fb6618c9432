#!/bin/bash
# A/B of the stored order of a block's positions: basis order (LPP_PB_PERM=0) against the order of the in-block list lengths (default)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for w in 0 1 0 1; do
  rm -rf $R/gpurun_out/prof_ab
  LPP_PB_PERM=$w timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg > /tmp/ab.json 2>/dev/null
  echo "== LPP_PB_PERM=$w"; grep -E "k_pb_down|k_pb_up" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | sed 's/"void lpp::\(k_pb_[a-z]*\)\([^"]*\)"/\1\2/' | cut -c1-60,61- | awk -F, '{print $1,$(NF-5),$(NF-3)}'
  python3 -c "import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'],d['config']['coefficients_vs_cpu_oracle']['max_rel_diff'])"
done
rm -rf $R/gpurun_out/prof_ab
