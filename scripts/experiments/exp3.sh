#!/bin/bash
# classify gather traffic of the sliced kernel on C2 (diagnostic builds; results are wrong by design)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp; O=$R/gpurun_out/exp3; mkdir -p $O; cd /tmp
export LPP_SPMV_KERNEL=2 LPP_K2_VARIANT=0
for d in 1 2 3; do
  export LPP_K2_DEBUG=$d
  rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/d$d -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/d$d.log 2>&1
  python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('DBG$d spmv_ms', d['roofline']['spmv_ms'])"
done
unset LPP_K2_DEBUG
python3 - <<PY
import csv,glob,collections
for d in ['d1','d2','d3']:
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'][:30],r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k,c),v in sorted(agg.items()):
            if 'spmv' in k: print(d,k,c,len(v),sum(v)/len(v), 'GB=%.1f'%(sum(v)/len(v)*128/1e9))
PY
find $O -size +5M -delete
