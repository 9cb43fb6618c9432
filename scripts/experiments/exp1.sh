#!/bin/bash
# experiment matrix on C2
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['spmv_ms'], d['roofline']['achieved'], d['ms_per_step'])"; }
run LPP_SPMV_KERNEL=2
run LPP_SPMV_KERNEL=2 LPP_SPMV_BLOCKS=2048
run LPP_SPMV_KERNEL=2 LPP_SPMV_BLOCKS=1024
run LPP_SPMV_KERNEL=1 LPP_SPMV_G=8
run LPP_SPMV_KERNEL=1 LPP_SPMV_G=32
run LPP_SPMV_KERNEL=1 LPP_SPMV_G=16 LPP_SPMV_BLOCKS=2048
run LPP_SPMV_KERNEL=1 LPP_SPMV_G=16 LPP_SPMV_BLOCKS=1024
