#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or linear or conv" > gpurun_out/r3_o_test.log 2>&1
rc=$?
tail -5 gpurun_out/r3_o_test.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
timeout -k 10 600 python tools/gemm_probe.py fc1 qkv fc2 out conv1 conv2 > gpurun_out/r3_o_probe.log 2>&1
grep -E "default|8ph|persist 160x128 |persist 256x128" gpurun_out/r3_o_probe.log
