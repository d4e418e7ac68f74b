#!/bin/bash
# The driver's N > 1 launch line, rehearsed on a one-GPU box (see bench.py, W2VS_REHEARSE_ONE_GPU): 2 ranks, both on cuda:0, gloo.
# bash tools/rehearse_multirank.sh [wire]   ->  gpurun_out/rehearse_2rank.json
set -e
mkdir -p gpurun_out
export W2VS_REHEARSE_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 6 --warmup 2 --wire ${1:-fp32} > gpurun_out/rehearse_2rank.out 2> gpurun_out/rehearse_2rank.err
grep -c '^{"metric"' gpurun_out/rehearse_2rank.out
grep '^{"metric"' gpurun_out/rehearse_2rank.out | tail -1 > gpurun_out/rehearse_2rank.json
cut -c1-400 gpurun_out/rehearse_2rank.json
