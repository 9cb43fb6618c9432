#!/bin/bash
# BASELINE config 5's lattice as the driver's 8-GPU node would run it (bench.py --gpus N --engine onthefly --workload
# hubbard_4x5_7up6down_pbc_U4), rehearsed with N ranks sharing ONE GPU over gloo: functional check of the line + per-rank memory.
mkdir -p gpurun_out
export LPP_BENCH_BACKEND=gloo
N=${1:-4}
timeout -k 10 ${2:-900} python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N --steps 2 --warmup 1 --engine onthefly --workload hubbard_4x5_7up6down_pbc_U4 --no-cpu-baseline > gpurun_out/reh_c5_$N.log 2>&1 || { tail -20 gpurun_out/reh_c5_$N.log; exit 1; }
tail -1 gpurun_out/reh_c5_$N.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n', d['n_gpus'], d['config']['engine'], d['config']['exchange'], 'rows', d['config']['rows'], 'it/s %.4f' % d['value'], 'e0', d['e0_after_steps'], 'per-rank GB', d['config']['per_rank_memory_GB'], 'device GB', d['config']['device_memory_after_setup_GB'])"
