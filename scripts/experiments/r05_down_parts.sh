#!/bin/bash
# Timing-only decomposition of the chained coupling kernel k_pb_down<RMW,PF> at config 2: scratch libraries (LIBS="tag tag ..",
# scripts/experiments/_build/liblpp_engine_<tag>.so, each built with one part of the kernel compiled out -- results wrong by
# construction) under the kernel trace; prints the average duration of the two kernels of the step.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for tag in ${LIBS:-base}; do
  rm -rf $R/gpurun_out/prof_dp
  export LPP_ENGINE_LIB=$R/scripts/experiments/_build/liblpp_engine_$tag.so
  [ "$tag" = intree ] && unset LPP_ENGINE_LIB
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dp -- python3 $R/scripts/experiments/r05_step_time.py ${WL:-hubbard_4x4_half_filling_pbc_U4} 20 > $R/gpurun_out/dp.out 2> $R/gpurun_out/dp.err || { tail -5 $R/gpurun_out/dp.err; exit 1; }
  echo "== $tag: $(tail -1 $R/gpurun_out/dp.out)"
  python3 - "$R"/gpurun_out/prof_dp/*/*kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_pb_" in r["Name"] and "diag" not in r["Name"]:
        print("   %-60s %4s calls  %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
rm -rf $R/gpurun_out/prof_dp
