#!/bin/bash
# `python bench.py --gpus 2` with NO launcher around it, rehearsed on a one-GPU box (see bench.py: self_launch,
# W2VS_REHEARSE_ONE_GPU): bench.py itself starts 2 ranks through torch.distributed.run, both on cuda:0, gloo.
# bash tools/rehearse_multirank.sh [wire]   ->  gpurun_out/rehearse_2rank.json
set -e
mkdir -p gpurun_out
export W2VS_REHEARSE_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
unset WORLD_SIZE RANK LOCAL_RANK
timeout -k 10 500 python bench.py --gpus 2 --steps 6 --warmup 2 --wire ${1:-bf16} > gpurun_out/rehearse_2rank.out 2> gpurun_out/rehearse_2rank.err
grep -c '^{"metric"' gpurun_out/rehearse_2rank.out
grep '^{"metric"' gpurun_out/rehearse_2rank.out | tail -1 > gpurun_out/rehearse_2rank.json
cut -c1-400 gpurun_out/rehearse_2rank.json
# and without the rehearsal switch a one-GPU box must REFUSE --gpus 2 (exit code != 0, no result line)
unset W2VS_REHEARSE_ONE_GPU
if python bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/refuse_2rank.out 2> gpurun_out/refuse_2rank.err; then echo "NOT REFUSED"; exit 1; fi
echo "refused: $(tail -1 gpurun_out/refuse_2rank.err)"
