#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention or attn or dropout" > gpurun_out/r3_h_test.log 2>&1
rc=$?
tail -4 gpurun_out/r3_h_test.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out"; exit 1; fi
timeout -k 10 300 python tools/attn_probe.py > gpurun_out/r3_h_probe.log 2>&1
cat gpurun_out/r3_h_probe.log
