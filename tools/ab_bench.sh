#!/bin/bash
# Same-box A/B of two builds of the library in the step: bench.py alternately with W2VS_LIB = $1 and $2, $3 rounds (default 3).
#   bash tools/ab_bench.sh wav2vec-s_amd/libw2vs_prev.so wav2vec-s_amd/libw2vs.so 3 [extra bench.py flags]
A=$(readlink -f $1); B=$(readlink -f $2); N=${3:-3}; shift 3
for i in $(seq $N); do
  for L in $A $B; do
    W2VS_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --no-gemm-peak "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('%-40s mean %.3f median %.3f ms  nt %.0f tn %.0f TF/s' % ('$(basename $L)', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['all_gemm_nt_tflops'], d['roofline']['all_gemm_tn_tflops']))"
  done
done
