#!/bin/bash
# round 3, step A: correctness of the 8-phase NT kernel, then the tile probe
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest "tests/test_kernels_gpu.py::test_gemm_nt_every_variant[8-256]" "tests/test_kernels_gpu.py::test_gemm_nt_every_variant[8-320]" -x -q > gpurun_out/r3_a_test.log 2>&1
rc=$?
tail -5 gpurun_out/r3_a_test.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "test timed out, stopping"; exit 1; fi
timeout -k 10 420 python tools/gemm_probe.py fc1 qkv conv1 conv2 sq4k > gpurun_out/r3_a_probe.log 2>&1
tail -80 gpurun_out/r3_a_probe.log
