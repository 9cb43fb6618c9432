#!/bin/bash
# Any counter sets of one bench configuration, per kernel (separate --pmc passes, no trace options).
# usage: SETS="A_sum B_sum|C_sum" ARGS="--workload ..." FILTER="k_pb_down" bash scripts/experiments/r05_counters.sh <tag>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-cnt}; export TMPDIR=/tmp; O=$R/gpurun_out/cnt_$TAG; mkdir -p $O; cd /tmp
CMDP="python3 $R/bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-e0-check --no-reortho-leg --no-generic-csr $ARGS"
i=0
IFS='|' read -ra LIST <<< "$SETS"
for set in "${LIST[@]}"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $CMDP > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
python3 - <<PY
import csv,glob,collections
pm=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        pm[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(pm):
    if not any(x in k for x in '${FILTER:-k_}'.split('|')): continue
    print(k)
    for c in sorted(pm[k]):
        v=pm[k][c]; print('   %-42s %14.4g  (%d)' % (c, sum(v)/len(v), len(v)))
PY
rm -rf $O/p[0-9]*
