#!/bin/bash
# split-panel layout at config 2: grid size of the sliced launches, number of parts -- time and fabric reads of the out-part kernels
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/prof_gen
for cfg in "LPP_SPMV_BLOCKS=2048" "LPP_SPMV_BLOCKS=2048 LPP_SPLIT_PARTS=1" "LPP_SPMV_BLOCKS=1024" "LPP_SPMV_BLOCKS=512 LPP_SPLIT_PARTS=1"; do
  rm -rf $O; mkdir -p $O
  env $cfg ITERS=4 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/scripts/experiments/r03_generic_leg.py > $O/trace.log 2>&1
  env $cfg ITERS=4 timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_HIT_sum --output-format csv -d $O/ea -- python3 $R/scripts/experiments/r03_generic_leg.py > $O/ea.log 2>&1
  echo "== $cfg"; grep -E "k_spmv" $O/trace/*/*kernel_stats.csv | cut -c1-140
  python3 - <<PY
import csv,glob,collections
pm=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/ea/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_spmv' in r['Kernel_Name']: pm[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in pm:
    c={n:sum(v)/len(v) for n,v in pm[k].items()}
    print('  ',k, 'read GB %.2f'%(c.get('TCC_EA0_RDREQ_128B_sum',0)*128/1e9), 'hits %.3g'%c.get('TCC_HIT_sum',0))
PY
done
rm -rf $O
