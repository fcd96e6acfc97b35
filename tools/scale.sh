#!/bin/bash
# The exact command lines of the driver's 1 / 2 / 4 / 8-GPU scaling run of bench.py on ONE node (one rank per GPU, RCCL over xGMI),
# TWICE per N:
#   default      rank-local BatchNorm statistics in the PAFPN (SURVEY 8e option (i): "all-reduce for gradients only", north_star) --
#                the line the driver's `bench.py --gpus N` produces, and the one the >= 70 % 1 -> 8 efficiency target refers to
#   --sync-bn    the reference's own DDP semantics (train.py:167 Trainer(sync_batchnorm=True)): 56 small statistics all-reduces per
#                step captured into the hipGraphs (independent units share one); expected BELOW 70 % (56 dependent collectives x
#                20-30 us on a 4.8 ms step)
# usage: bash tools/scale.sh [steps] [warmup] > scale.jsonl      (needs an N-GPU node; nothing here can run on the 1-GPU boxes)
steps=${1:-100}; warmup=${2:-20}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus 1 --steps $steps --warmup $warmup
for n in 2 4 8; do
  for mode in "" "--sync-bn"; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --steps $steps --warmup $warmup $mode
  done
done
# every line carries config.sync_batchnorm (which semantics ran), config.collective_ranks_verified (an all-reduce of ones across the
# ranks, = n), config.allreduce_exposed_ms (what the step waits for the last gradient bucket after its own backward),
# config.allreduce_update_ms_per_bucket and, with --sync-bn, config.sync_batchnorm_collectives_per_step
