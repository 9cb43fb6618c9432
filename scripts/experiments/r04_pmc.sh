#!/bin/bash
# Counter passes (fabric requests, L2 hits, SQ waits / instruction mix) of one bench configuration, summarised per kernel.
# PASSES="ea write sq sq2" picks the counter passes.
# usage: ENVS="LPP_PB_SEG=1" ARGS="--engine onthefly --workload hubbard_4x5_7up6down_pbc_U4" FILTER="k_pb_" bash scripts/experiments/r04_pmc.sh <tag>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-pmc}; export TMPDIR=/tmp; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O; cd /tmp
for e in $ENVS; do export $e; done
CMDP="python3 $R/bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-e0-check --no-reortho-leg ${GENERIC:---no-generic-csr} $ARGS"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMDP > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
PASSES=${PASSES:-"ea write sq sq2"}
[[ " $PASSES " == *" ea "* ]] && timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum --output-format csv -d $O/ea -- $CMDP > $O/ea.log 2>&1
[[ " $PASSES " == *" write "* ]] && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMDP > $O/write.log 2>&1
[[ " $PASSES " == *" sq "* ]] && timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq -- $CMDP > $O/sq.log 2>&1
[[ " $PASSES " == *" sq2 "* ]] && timeout -k 10 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2 -- $CMDP > $O/sq2.log 2>&1
python3 - <<PY
import csv,glob,collections
for f in glob.glob('$O/trace/*/*_kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    with open('$O/kernel_stats.csv','w') as g:
        w=csv.writer(g); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','Percentage','MinNs','MaxNs'])
        for r in rows: w.writerow([r['Name'][:120],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['Percentage'],r['MinNs'],r['MaxNs']])
            
pm=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ['ea','write','sq','sq2']:
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            pm[r['Kernel_Name'][:100]][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$O/pmc_summary.csv','w') as g:
    w=csv.writer(g); w.writerow(['Kernel','Counter','Dispatches','MeanPerDispatch'])
    for k in sorted(pm):
        for c in sorted(pm[k]):
            v=pm[k][c]; w.writerow([k,c,len(v),sum(v)/len(v)])
PY
rm -rf $O/trace $O/ea $O/write $O/sq $O/sq2
grep -E "${FILTER:-k_}" $O/kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
grep -E "${FILTER:-k_}" $O/pmc_summary.csv | cut -c1-200
