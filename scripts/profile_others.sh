#!/bin/bash
# round profiles of the other BASELINE configurations and of the matrix-free engine
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
BENCH_ARGS="--workload heisenberg_chain_L28_sz0_obc" bash scripts/profile_round.sh ${1:-r01h}_c3 > gpurun_out/po_c3.out 2>&1
echo c3 done
BENCH_ARGS="--workload tj_4x5_9up9down_complex" bash scripts/profile_round.sh ${1:-r01h}_c4 > gpurun_out/po_c4.out 2>&1
echo c4 done
BENCH_ARGS="--engine onthefly" bash scripts/profile_round.sh ${1:-r01h}_c2otf > gpurun_out/po_otf.out 2>&1
echo otf done
LPP_ONTHEFLY_KRON=1 BENCH_ARGS="--engine onthefly --no-cpu-baseline --no-e0-check" bash scripts/profile_round.sh ${1:-r01h}_c2otf_kron > gpurun_out/po_otfk.out 2>&1
echo otf kron done
BENCH_ARGS="--workload hubbard_chain_L12_half_filling_U4" bash scripts/profile_round.sh ${1:-r01h}_c1 > gpurun_out/po_c1.out 2>&1
echo c1 done
for t in c3 c4 c2otf c2otf_kron c1; do echo == $t; cut -c1-260 gpurun_out/profile_${1:-r01h}_$t/bench.json; echo; grep -E "spmv" gpurun_out/profile_${1:-r01h}_$t/kernel_stats.csv | cut -c1-150 | head -2; grep -E "spmv" gpurun_out/profile_${1:-r01h}_$t/pmc_summary.csv | cut -c1-200 | head -6; done
