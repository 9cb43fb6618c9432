#!/bin/bash
# Plain 12-byte CSR leg at config 2: r03_generic_prof.sh (kernel trace, fabric reads / writes) plus the texture-addresser / vector-L1 / translation
# counters of the same command (separate --pmc passes).
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
NAME=${NAME:-hubbard_4x4_half_filling_pbc_U4} bash $R/scripts/experiments/r03_generic_prof.sh
O=$R/gpurun_out/prof_gen2; rm -rf $O; mkdir -p $O
CMD="python3 $R/scripts/experiments/r03_generic_leg.py ${NAME:-hubbard_4x4_half_filling_pbc_U4}"
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_LFIFO_FULL_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $CMD > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
python3 - <<PY
import csv,glob,collections
pm=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_spmv' in r['Kernel_Name']: pm[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(pm):
    print(k)
    for c in sorted(pm[k]):
        v=pm[k][c]; print('   %-42s %14.4g  (%d)' % (c, sum(v)/len(v), len(v)))
PY
rm -rf $O
