#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "layernorm or ln or encoder_layer" 2>&1 | tail -2
for v in 0 1; do echo -n "lean=$v "; W2VS_LN_LEAN=$v timeout -k 10 100 python tools/bench_kernels.py "ln_fwd" 2>&1 | tail -1; done
for i in 1 2 3; do
  for v in 0 1; do
    W2VS_LN_LEAN=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --no-gemm-peak 2>/dev/null > gpurun_out/ab.json || exit 1
    python - "lean=$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
print(sys.argv[1], d['ms_per_step'], d['ms_per_step_median'])
PY
  done
done
