#!/bin/bash
# A/B of the wave-specialised chained in-block kernel (k_pb_up_ws): loader waves / pairs per lane / chunk of r loads.
# Rebuilds lpp_pb.o on the box per variant (kernel trace only).  usage: bash scripts/experiments/r03_ws_ab.sh
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() {
  cd /tmp; rm -rf $R/gpurun_out/prof_ab
  env $1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-generic-csr --no-e0-check --no-reortho-leg > /tmp/ab.json 2>/dev/null
  echo "== $2 $1"; grep -E "k_pb_down|k_pb_up" $R/gpurun_out/prof_ab/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
  python3 -c "import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'])"
}
run LPP_PB_WS=0 "k_pb_up<CHAIN>"
for v in "5 21 3 0" "5 21 3 2" "6 18 3 2"; do
  set -- $v
  cd $R/lanczosplusplus_amd/csrc && rm -f lpp_pb.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -Wno-unused-result --offload-arch=gfx950 -I../../include -DLPP_PBWS_LOADERS=$1 -DLPP_PBWS_NQ=$2 -DLPP_PBWS_CH=$3 -DLPP_PBWS_AUX=$4" liblpp_engine.so > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; continue; }
  run "LPP_PB_WS=1 LPP_PB_WS_BETA=1" "loaders=$1 nq=$2 ch=$3 aux=$4"
  run "LPP_PB_WS=1 LPP_PB_WS_BETA=0" "loaders=$1 nq=$2 ch=$3 aux=$4"
done
rm -rf $R/gpurun_out/prof_ab
