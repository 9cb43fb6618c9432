#!/bin/bash
# (7,6) sector of the 4x5 lattice: in-block kernel with the per-position template (LPP_PB_SEG=0: k_pb_up_big2) against the
# decomposition by the high sites (k_pb_up_seg); kernel trace only.  CFGS="ENV=.. ENV=.." overrides the list.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
WL=${WL:-hubbard_4x5_7up6down_pbc_U4}
for cfg in ${CFGS:-"LPP_PB_SEG=0" "LPP_PB_SEG=1"}; do
  rm -rf $R/gpurun_out/prof_ab
  env $cfg LPP_VERBOSE=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --engine onthefly --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg > $R/gpurun_out/ab.json 2> $R/gpurun_out/ab.err || { tail -5 $R/gpurun_out/ab.err; exit 1; }
  echo "== $cfg"; grep "lpp:" $R/gpurun_out/ab.err | head -5; python3 -c "
import json;j=json.loads(open('$R/gpurun_out/ab.json').read().strip().splitlines()[-1]);print(j['value'],j['ms_per_step'],j['roofline'].get('spmv_ms'),j['config'].get('layout'))"
  python3 - "$R"/gpurun_out/prof_ab/*/*kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if ("k_pb_" in r["Name"] or "k_axpy" in r["Name"] or "k_spmv" in r["Name"]) and "diag" not in r["Name"]:
        print("   %-70s %4s calls  %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
rm -rf $R/gpurun_out/prof_ab
