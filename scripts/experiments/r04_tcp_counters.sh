#!/bin/bash
# Texture-addresser / vector-L1 counters of one bench configuration (what the memory pipeline of a CU is busy with), per kernel.
# usage: ARGS="--workload ..." FILTER="k_pb_" bash scripts/experiments/r04_tcp_counters.sh <tag>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-tcp}; export TMPDIR=/tmp; O=$R/gpurun_out/tcp_$TAG; mkdir -p $O; cd /tmp
CMDP="python3 $R/bench.py --steps ${STEPS:-6} --warmup 1 --no-cpu-baseline --no-e0-check --no-reortho-leg --no-generic-csr $ARGS"
i=0
for set in "TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
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
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
