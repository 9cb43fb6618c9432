#!/bin/bash
# bench.py multi-rank path rehearsed on ONE GPU: ranks share cuda:0, collectives over gloo (RCCL needs a GPU per rank)
mkdir -p gpurun_out
export LPP_BENCH_BACKEND=gloo
W=${1:-hubbard_4x4_half_filling_pbc_U4}
for n in 2 4; do
  for eng in stored onthefly; do
    timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 10 --warmup 3 --workload $W --engine $eng > gpurun_out/reh_${n}_$eng.log 2>&1 || { tail -20 gpurun_out/reh_${n}_$eng.log; exit 1; }
    tail -1 gpurun_out/reh_${n}_$eng.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n', d['n_gpus'], d['config']['engine'], d['config']['exchange'], 'it/s %.2f' % d['value'], 'e0', d['e0_after_steps'], d['config'].get('layout'))"
  done
done
