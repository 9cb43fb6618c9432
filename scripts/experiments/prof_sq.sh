#!/bin/bash
# SQ / TCP counters for kernels matching $KPAT of the bench configuration in $BENCH_ARGS
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp
TAG=${1:-sq}
KPAT=${KPAT:-spmv}
O=$R/gpurun_out/profsq_$TAG
mkdir -p $O; cd /tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/sq3 -- $B > $O/sq3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $O/sq4 -- $B > $O/sq4.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ['sq','sq2','sq3','sq4']:
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'][:34],r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k,c),v in sorted(agg.items()):
            if '$KPAT' in k: print(d,k,c,len(v),'%.4g'%(sum(v)/len(v)))
PY
find $O -size +5M -delete
