#!/bin/bash
# kernel times (rocprofv3 --kernel-trace --stats) + L2-miss read requests (separate --pmc pass) of one bench configuration:
#   BENCH_ARGS="--workload ... --engine ..." bash scripts/experiments/prof_traffic.sh <tag>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp
TAG=${1:-traffic}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O; cd /tmp
B="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline $BENCH_ARGS"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum --output-format csv -d $O/pmc -- $B > $O/pmc.log 2>&1
python3 - <<PY
import csv,glob,collections
for f in glob.glob('$O/kt/*/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
for f in glob.glob('$O/pmc/*/*_counter_collection.csv'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:40],r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(agg.items()):
        if "spmv" in k or "kron" in k: print(k,c,len(v),'%.5g'%(sum(v)/len(v)))
PY
find $O -size +5M -delete
