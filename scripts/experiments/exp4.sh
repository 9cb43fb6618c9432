#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['spmv_ms'], d['roofline']['achieved'], d['ms_per_step'], d['e0_after_steps'])"; }
for v in 0 2 4 6; do run LPP_SPMV_KERNEL=3 LPP_K2_VARIANT=$v; done
for v in 0 2 4 6; do run LPP_SPMV_KERNEL=2 LPP_K2_VARIANT=$v; done
