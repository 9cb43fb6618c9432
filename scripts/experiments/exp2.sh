#!/bin/bash
# K2 variants on C2: bit0 NT, bit1 XCD map, bit2 U=8
run() { echo "== $*"; env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['spmv_ms'], d['roofline']['achieved'], d['ms_per_step'], d['e0_after_steps'])"; }
for v in 0 1 2 3 4 5 6 7; do run LPP_SPMV_KERNEL=2 LPP_K2_VARIANT=$v; done
run LPP_SPMV_KERNEL=2 LPP_K2_VARIANT=3 LPP_SPMV_BLOCKS=2048
run LPP_SPMV_KERNEL=2 LPP_K2_VARIANT=7 LPP_SPMV_BLOCKS=2048
