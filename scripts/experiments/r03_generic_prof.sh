#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/prof_gen; rm -rf $O; mkdir -p $O
CMD="python3 $R/scripts/experiments/r03_generic_leg.py $NAME"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum --output-format csv -d $O/ea -- $CMD > $O/ea.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/wr -- $CMD > $O/wr.log 2>&1
grep -E "k_spmv" $O/trace/*/*kernel_stats.csv | cut -c1-260
python3 - <<PY
import csv,glob,collections
pm=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ['ea','wr']:
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'k_spmv' in r['Kernel_Name']: pm[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in pm:
    c={n:sum(v)/len(v) for n,v in pm[k].items()}
    rd=c.get('TCC_EA0_RDREQ_128B_sum',0)*128+c.get('TCC_EA0_RDREQ_64B_sum',0)*64+c.get('TCC_EA0_RDREQ_32B_sum',0)*32
    print(k, 'read GB %.2f'%(rd/1e9), 'write GB %.2f'%(c.get('WRITE_SIZE',0)*1024/1e9), 'hits %.3g'%c.get('TCC_HIT_sum',0))
PY
rm -rf $O/trace $O/ea $O/wr
