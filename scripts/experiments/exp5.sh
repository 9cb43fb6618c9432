#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $WL 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['rows'], d['config']['nnz'], 'spmv_ms', d['roofline']['spmv_ms'], 'GB/s', d['roofline']['achieved'], 'ms/step', d['ms_per_step'], d['e0_after_steps'], 'asm_s', d['config']['assembly_s'])"; }
for WL in heisenberg_chain_L28_sz0_obc tj_4x5_9up9down_complex hubbard_chain_L12_half_filling_U4; do
  export WL
  for k in 1 2 3; do run LPP_SPMV_KERNEL=$k; done
done
