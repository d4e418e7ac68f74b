#!/bin/bash
# attention tests, then tools/attn_probe.py alternately on libw2vs_prev.so and libw2vs.so (same box)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attn or attention" > gpurun_out/attn_tests.log 2>&1 || { tail -30 gpurun_out/attn_tests.log; exit 1; }
tail -1 gpurun_out/attn_tests.log
for i in 1 2; do for L in libw2vs_prev.so libw2vs.so; do echo "== $L"; W2VS_LIB=$PWD/wav2vec-s_amd/$L timeout -k 10 120 python tools/attn_probe.py 2>&1 | grep -E "cfgB m16 r8|large|dense N=818 \(m=818,r=0\) p0.1"; done; done
