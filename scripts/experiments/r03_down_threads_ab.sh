#!/bin/bash
# A/B of the coupling kernel's workgroup size in the chained step (LPP_PB_DOWN_THREADS): 16 or 8 waves per CU -- the waves' gathers
# in flight (16 x 3 chunks x 4 x 8 lines = 1536) are six times the 256 lines of L1, and L1 hits are worth a third of the kernel
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for w in 1024 512 1024 512; do
  rm -rf $R/gpurun_out/prof_ab
  LPP_PB_DOWN_THREADS=$w timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg > /tmp/ab.json 2>/dev/null
  echo "== threads $w"; grep -E "k_pb_down|k_pb_up" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | sed 's/"void lpp::\(k_pb_[a-z]*\)\([^"]*\)"/\1\2/' | awk -F, '{print $1,$(NF-5),$(NF-4)}'
  python3 -c "import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'],d['config']['coefficients_vs_cpu_oracle']['max_rel_diff'])"
done
rm -rf $R/gpurun_out/prof_ab
