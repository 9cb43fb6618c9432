#!/bin/bash
# A/B of engine builds: every scripts/experiments/variants/lib_*.so is copied over the in-tree library (on the GPU box's copy of the
# tree) and the short bench is run under the kernel trace.  usage: bash scripts/experiments/ab_variants.sh [bench args]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp
O=$R/gpurun_out/ab_var
mkdir -p $O
cp $R/lanczosplusplus_amd/csrc/liblpp_engine.so $O/orig.so
cd /tmp
for lib in $R/scripts/experiments/variants/lib_*.so; do
  v=$(basename $lib .so)
  cp $lib $R/lanczosplusplus_amd/csrc/liblpp_engine.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$v -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-generic-csr --no-e0-check "$@" > $O/$v.json 2> $O/$v.err
  echo "== $v"
  python3 - <<PY
import csv,glob,json
for f in glob.glob('$O/t_$v/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('k_pb_up','k_pb_down','k_axpy_nrm','k_spmv','k_pb_combine')):
            print('   %-60s calls %3s avg %9.1f us  min %9.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
try:
    d=json.load(open('$O/$v.json')); print('   it/s %.1f  ms/step %.3f  e0 %.9f' % (d['value'], d['ms_per_step'], d['e0_after_steps']))
except Exception as ex: print('   bench line missing', ex)
PY
  rm -rf $O/t_$v
done
cp $O/orig.so $R/lanczosplusplus_amd/csrc/liblpp_engine.so
