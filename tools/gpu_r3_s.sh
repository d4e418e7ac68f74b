#!/bin/bash
# interleaved A/B of two library builds on the headline bench (mean / median ms per step)
mkdir -p gpurun_out
for i in 1 2 3; do
  for lib in libw2vs.so libw2vs_nt.so; do
    W2VS_LIB=$PWD/wav2vec-s_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --no-gemm-peak 2>/dev/null > gpurun_out/ab.json || exit 1
    python - "$lib" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
print(sys.argv[1], d['ms_per_step'], d['ms_per_step_median'])
PY
  done
done
