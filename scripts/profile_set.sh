#!/bin/bash
# The round's profile set in two GPU calls, so that no kept bench line says "stale":
#   call 1:  bash scripts/profile_set.sh <tag> counters   kernel trace + PMC passes of every configuration (profile_round.sh),
#            plus the kernel trace of a reorthogonalised run on the pitched product-basis vectors (config 2, 22 steps)
#   here:    bash scripts/stamp_round.sh <tag>            copies the summaries to profiles/ and stamps profiles/traffic.json
#   call 2:  bash scripts/profile_set.sh <tag> lines      re-runs ONLY the bench lines (same sources, traffic.json now matches)
#   here:    bash scripts/stamp_round.sh <tag> lines      copies the lines over the ones of call 1
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05}; MODE=${2:-counters}
cd $R; mkdir -p gpurun_out
cfg() { # name, bench args, env
  case $1 in
    c2_stored) echo "" ;;
    c2otf) echo "--engine onthefly" ;;
    c2otf_kron) echo "--engine onthefly --no-generic-csr" ;; # with LPP_ONTHEFLY_KRON=1: the fused block-order kernel
    c3) echo "--workload heisenberg_chain_L28_sz0_obc" ;;
    c4) echo "--workload tj_4x5_9up9down_complex" ;;
    c1) echo "--workload hubbard_chain_L12_half_filling_U4" ;;
    c5_76) echo "--engine onthefly --workload hubbard_4x5_7up6down_pbc_U4 --no-generic-csr" ;;
    cx87) echo "--workload hubbard_4x4_8up7down_complex_U4 --no-generic-csr" ;; # complex hoppings beyond one LDS window (pieces form, four value groups)
  esac
}
for c in ${CONFIGS:-c2_stored c2otf c2otf_kron c3 c4 c1 c5_76}; do
  if [ $c = c2otf_kron ]; then export LPP_ONTHEFLY_KRON=1; else unset LPP_ONTHEFLY_KRON; fi
  if [ $MODE = counters ]; then
    SQ_PASS=$([ $c = c2_stored ] && echo 1) BENCH_ARGS="$(cfg $c)" bash scripts/profile_round.sh ${T}_$c > gpurun_out/ps_$c.out 2>&1 || { tail -5 gpurun_out/ps_$c.out; exit 1; }
  else
    mkdir -p gpurun_out/profile_${T}_$c
    (cd /tmp && timeout -k 10 500 python3 $R/bench.py --steps 40 --warmup 5 $(cfg $c) > $R/gpurun_out/profile_${T}_$c/bench.json 2> $R/gpurun_out/profile_${T}_$c/bench.err) || { tail -5 gpurun_out/profile_${T}_$c/bench.err; exit 1; }
  fi
  echo "$c done: $(python3 -c "import json;d=json.loads(open('gpurun_out/profile_${T}_$c/bench.json').read().strip().splitlines()[-1]);print(round(d['value'],1),'it/s; traffic',d['roofline'].get('traffic'),d['roofline'].get('traffic_note',''))")"
done
if [ $MODE = counters ] && [ -z "$NO_REORTHO" ]; then
  export TMPDIR=/tmp; O=$R/gpurun_out/profile_${T}_reortho; mkdir -p $O; cd /tmp
  CMDR="python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-generic-csr --no-e0-check"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMDR > $O/bench.json 2> $O/trace.log
  python3 - <<PY
import csv,glob
for f in glob.glob('$O/trace/*/*_kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    with open('$O/kernel_stats.csv','w') as g:
        w=csv.writer(g); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','Percentage','MinNs','MaxNs'])
        for r in rows: w.writerow([r['Name'][:120],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['Percentage'],r['MinNs'],r['MaxNs']])
PY
  rm -rf $O/trace; head -8 $O/kernel_stats.csv
fi
