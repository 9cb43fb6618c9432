#!/bin/bash
# traffic of the current default SpMV kernel on a workload: exact EA read sizes + write size + kernel trace
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp
TAG=${1:-default}
O=$R/gpurun_out/prof3_$TAG
mkdir -p $O
cd /tmp
B="python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
[ -n "$LPP_WL" ] && B="$B --workload $LPP_WL"
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum --output-format csv -d $O/ea -- $B > $O/ea.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/wr -- $B > $O/wr.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fs -- $B > $O/fs.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1
find $O -size +5M -delete
python3 - <<PY
import csv,glob,collections
for d in ['ea','wr','fs']:
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'][:34],r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k,c),v in sorted(agg.items()):
            if 'spmv' in k: print(d,k,c,len(v),sum(v)/len(v))
for f in glob.glob('$O/trace/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'spmv' in r['Name'] or 'axpy' in r['Name'] or 'swap' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
