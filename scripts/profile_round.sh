#!/bin/bash
# Round profile: the default bench line + rocprofv3 kernel trace and PMC passes of the SAME command.
# usage: [BENCH_ARGS="--workload ... --engine ..."] bash scripts/profile_round.sh r01   (outputs under gpurun_out/profile_<tag>/)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-r01}
export TMPDIR=/tmp
O=$R/gpurun_out/profile_$TAG
mkdir -p $O
cd /tmp
CMD="python3 $R/bench.py --steps 40 --warmup 5 $BENCH_ARGS"
$CMD > $O/bench.json 2> $O/bench.err
CMDP="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg $BENCH_ARGS"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMDP > $O/trace.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMDP > $O/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMDP > $O/write.log 2>&1
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum --output-format csv -d $O/ea -- $CMDP > $O/ea.log 2>&1
if [ -n "$SQ_PASS" ]; then
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq -- $CMDP > $O/sq.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2 -- $CMDP > $O/sq2.log 2>&1
fi
python3 - <<PY
import csv,glob,collections,json
for f in glob.glob('$O/trace/*/*_kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    with open('$O/kernel_stats.csv','w') as g:
        w=csv.writer(g); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','Percentage','MinNs','MaxNs'])
        for r in rows: w.writerow([r['Name'][:120],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['Percentage'],r['MinNs'],r['MaxNs']])
pm=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ['fetch','write','ea','sq','sq2']:
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            pm[r['Kernel_Name'][:100]][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$O/pmc_summary.csv','w') as g:
    w=csv.writer(g); w.writerow(['Kernel','Counter','Dispatches','MeanPerDispatch'])
    for k in sorted(pm):
        for c in sorted(pm[k]):
            v=pm[k][c]; w.writerow([k,c,len(v),sum(v)/len(v)])
PY
rm -rf $O/trace $O/fetch $O/write $O/ea $O/sq $O/sq2
cat $O/bench.json; head -8 $O/kernel_stats.csv; grep -E "spmv|k_pb|k_tj" $O/pmc_summary.csv || true
