#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest "tests/test_kernels_gpu.py::test_gemm_tn_group_equals_single_launches" -x -q > gpurun_out/r3_d_test.log 2>&1
rc=$?
tail -5 gpurun_out/r3_d_test.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "test timed out, stopping"; exit 1; fi
for r in 6544 5584 4800; do
W2VS_TN8=0 timeout -k 10 120 python tools/wgrad_group_probe.py $r || exit 1
W2VS_TN8=1 timeout -k 10 120 python tools/wgrad_group_probe.py $r || exit 1
done
W2VS_TN8=1 W2VS_TN8_DBG=1 timeout -k 10 120 python tools/wgrad_group_probe.py 6544
W2VS_TN8=1 W2VS_TN8_DBG=3 timeout -k 10 120 python tools/wgrad_group_probe.py 6544
