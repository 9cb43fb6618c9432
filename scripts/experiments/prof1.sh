#!/bin/bash
# rocprofv3 passes on the C2 bench (kernel trace + separate PMC passes)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TMPDIR=/tmp
K=${LPP_SPMV_KERNEL:-2}
export LPP_SPMV_KERNEL=$K
O=$R/gpurun_out/prof_k$K
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/tcc.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $O/tcp -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/tcp.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/sq.log 2>&1
find $O -name "*.csv" | head -50
# keep only small summaries (drop per-dispatch traces > 5 MB)
find $O -size +5M -delete
