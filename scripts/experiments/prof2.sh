#!/bin/bash
# exact HBM read request sizes + TLB + TA stalls for the SpMV kernel on C2
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp
K=${LPP_SPMV_KERNEL:-2}
export LPP_SPMV_KERNEL=$K
O=$R/gpurun_out/prof2_k$K
mkdir -p $O
cd /tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
[ -n "$LPP_WL" ] && B="$B --workload $LPP_WL"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/ea -- $B > $O/ea.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum TCC_READ_sum TCC_REQ_sum --output-format csv -d $O/ea2 -- $B > $O/ea2.log 2>&1
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --output-format csv -d $O/tlb -- $B > $O/tlb.log 2>&1
rocprofv3 --pmc TA_BUSY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d $O/ta -- $B > $O/ta.log 2>&1
rocprofv3 --pmc TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d $O/ta2 -- $B > $O/ta2.log 2>&1
find $O -size +5M -delete
python3 - <<PY
import csv,glob,collections
for d in ['ea','ea2','tlb','ta','ta2']:
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'][:36],r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k,c),v in sorted(agg.items()):
            if 'spmv' in k or 'axpy' in k: print(d,k,c,len(v),sum(v)/len(v))
PY
