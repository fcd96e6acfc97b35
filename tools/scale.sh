#!/bin/bash
# The exact command lines of the driver's 1 / 2 / 4 / 8-GPU scaling run of bench.py on ONE node (one rank per GPU, RCCL over xGMI).
# usage: bash tools/scale.sh [steps] [warmup] > scale.jsonl      (needs an N-GPU node; nothing here can run on the 1-GPU boxes)
steps=${1:-100}; warmup=${2:-20}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus 1 --steps $steps --warmup $warmup
for n in 2 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
    bench.py --gpus $n --steps $steps --warmup $warmup
done
# every line carries config.collective_ranks_verified (an all-reduce of ones across the ranks, = n), config.allreduce_exposed_ms
# (what the step waits for the last gradient bucket after its own backward) and config.allreduce_update_ms_per_bucket
