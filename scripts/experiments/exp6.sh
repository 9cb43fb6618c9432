#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['spmv_ms'], d['roofline']['achieved'], d['ms_per_step'])"; }
for v in 0 2 4 6; do run LPP_K2_VARIANT=$v; done
for v in 4 6; do run LPP_K2_VARIANT=$v LPP_SPMV_KERNEL=2; done
