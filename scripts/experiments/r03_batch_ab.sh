#!/bin/bash
# A/B of the number of template chunks whose LDS gathers are in flight together in k_pb_up (LPP_PB_BATCH, compile time): rebuilds on the box
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
for b in 2 3; do
  cd $R/lanczosplusplus_amd/csrc && rm -f lpp_pb.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wno-unused-result --offload-arch=gfx950 -I../../include -DLPP_PB_BATCH=$b" liblpp_engine.so > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; continue; }
  cd /tmp; rm -rf $R/gpurun_out/prof_ab
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg > /tmp/ab.json 2>/dev/null
  echo "== batch $b"; grep -E "k_pb_down|k_pb_up" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | sed 's/"void lpp::\(k_pb_[a-z]*\)[^"]*"/\1/' | cut -d, -f1-4
  python3 -c "import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'],d['config']['coefficients_vs_cpu_oracle']['max_rel_diff'])"
done
rm -rf $R/gpurun_out/prof_ab
