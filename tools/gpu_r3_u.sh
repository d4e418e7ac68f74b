#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_a_dist_gpu.py tests/test_kernels_gpu.py -q -x -k "not attention" > gpurun_out/r3_u_test.log 2>&1
rc=$?
tail -5 gpurun_out/r3_u_test.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
for i in 1 2; do
  for v in 0 1; do
    W2VS_PAIR_WGRADS=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --no-gemm-peak 2>/dev/null > gpurun_out/ab.json || exit 1
    python - "pair=$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
r=d['roofline']
print(sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], r['kernel'], r['achieved'], r['avg_launch_us'], r['all_gemm_tn_tflops'])
PY
  done
done
