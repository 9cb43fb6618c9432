#!/bin/bash
# A/B of the product-basis kernels: kernel-trace averages of k_pb_up / k_pb_down / k_axpy_nrm for a list of environment variants.
# usage: bash scripts/experiments/ab_pb.sh "VAR1=a VAR2=b" "VAR3=c" ...   (an empty string = defaults); output under gpurun_out/ab_pb/
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp
O=$R/gpurun_out/ab_pb
mkdir -p $O
cd /tmp
i=0
for v in "$@"; do
  i=$((i+1))
  ( export $v; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$i -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-generic-csr --no-e0-check $BENCH_ARGS > $O/run$i.json 2> $O/run$i.err )
  echo "== variant $i: [$v]"
  python3 - <<PY
import csv,glob,json
for f in glob.glob('$O/t$i/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('k_pb_up','k_pb_down','k_axpy_nrm','k_spmv','k_pb_combine')):
            print('   %-60s calls %3s avg %9.1f us  min %9.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
try:
    d=json.load(open('$O/run$i.json')); print('   it/s %.1f  ms/step %.3f  spmv_ms %.3f  e0 %.9f' % (d['value'], d['ms_per_step'], d['roofline']['spmv_ms'], d['e0_after_steps']))
except Exception as ex: print('   bench line missing', ex)
PY
  rm -rf $O/t$i
done
