#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['spmv_ms'], d['roofline']['achieved'], d['ms_per_step'])"; }
for v in 4 6 4 6; do run LPP_K2_VARIANT=$v LPP_BENCH_ENGINE=onthefly; done
for v in 4 6; do run LPP_K2_VARIANT=$v; done
for v in 4 6; do run LPP_K2_VARIANT=$v LPP_BENCH_WORKLOAD=heisenberg_chain_L28_sz0_obc; done
for v in 4 6; do run LPP_K2_VARIANT=$v LPP_BENCH_WORKLOAD=tj_4x5_9up9down_complex; done
