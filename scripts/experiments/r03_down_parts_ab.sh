#!/bin/bash
# Timing-only builds (WRONG results) of k_pb_down<RMW>: what each part costs -- the gathered lines (replaced by the block's own line:
# same instructions, L1 hits), the read of u, the store.  Rebuilds on the box; kernel trace only.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
[ -n "$ONLY" ] && VARS=($ONLY) || VARS=("" "-DLPP_PB_TIMING_DOWN_NTGATHER" "-DLPP_PB_TIMING_DOWN_OWNLINES" "-DLPP_PB_TIMING_DOWN_NOU" "-DLPP_PB_TIMING_DOWN_NOSTORE" "-DLPP_PB_TIMING_DOWN_OWNLINES@-DLPP_PB_TIMING_DOWN_NOU@-DLPP_PB_TIMING_DOWN_NOSTORE")
for d in "${VARS[@]}"; do
  d=${d//@/ }
  cd $R/lanczosplusplus_amd/csrc && rm -f lpp_pb.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wno-unused-result --offload-arch=gfx950 -I../../include $d" liblpp_engine.so > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; continue; }
  cd /tmp; rm -rf $R/gpurun_out/prof_ab
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg > /tmp/ab.json 2>/tmp/ab.err
  echo "== build '$d'"; grep -E "k_pb_down|k_pb_up" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | sed 's/"void lpp::\(k_pb_[a-z]*\)\([^"]*\)"/\1\2/' | awk -F, '{print $1,$(NF-5),$(NF-3)}'
done
rm -rf $R/gpurun_out/prof_ab
