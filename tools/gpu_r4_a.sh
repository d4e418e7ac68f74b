#!/bin/bash
# round-4 checkpoint: GPU suite, then conv-dgrad zero-block skip A/B on the bench (interleaved)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4_tests_b.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r4_tests_b.log
for i in 1 2; do
  for v in 0 1; do
    W2VS_CONV_DGRAD_SKIP=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --no-gemm-peak 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('skip=$v', d['ms_per_step'], d['ms_per_step_median'], d['roofline'].get('all_gemm_nt_tflops'))"
  done
done
