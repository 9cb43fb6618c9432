#!/bin/bash
# Timing-only builds (WRONG results) of k_pb_up<CHAIN>: what each part of it costs -- the re-read of r_{j-1}, the store of u, the LDS
# gathers, the template words, the axpy part of the staging.  Rebuilds on the box; kernel trace only.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
VARS=("" "-DLPP_PB_TIMING_NOYOLD" "-DLPP_PB_TIMING_NOSTORE" "-DLPP_PB_TIMING_NOGATHER" "-DLPP_PB_TIMING_NOWORDS" "-DLPP_PB_TIMING_NOSTAGE" "-DLPP_PB_TIMING_NOGATHER@-DLPP_PB_TIMING_NOWORDS" "-DLPP_PB_TIMING_NOGATHER@-DLPP_PB_TIMING_NOWORDS@-DLPP_PB_TIMING_NOSTORE@-DLPP_PB_TIMING_NOYOLD")
for d in "${VARS[@]}"; do
  d=${d//@/ }
  cd $R/lanczosplusplus_amd/csrc && rm -f lpp_pb.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wno-unused-result --offload-arch=gfx950 -I../../include $d" liblpp_engine.so > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; continue; }
  cd /tmp; rm -rf $R/gpurun_out/prof_ab
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg > /tmp/ab.json 2>/tmp/ab.err
  echo "== build '$d'"; grep -E "k_pb_down|k_pb_up" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | sed 's/"void lpp::\(k_pb_[a-z]*\)\([^"]*\)"/\1\2/' | awk -F, '{print $1,$(NF-5),$(NF-3)}'
done
rm -rf $R/gpurun_out/prof_ab
