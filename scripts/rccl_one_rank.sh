#!/bin/bash
# The data-parallel step on ONE GPU with a one-rank RCCL communicator (EVC_DP_FORCE=1): every collective of the
# step (per-group gradient all-reduces on the side streams, the MoE factor all-gathers, the student's own
# communicator, the loss all-reduce) goes through RCCL; the result must match the plain single-GPU run.
set -e
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
EVC_DP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps ${STEPS:-10} --warmup 3 --no_cpu_baseline
python bench.py --gpus 1 --steps ${STEPS:-10} --warmup 3 --no_cpu_baseline
