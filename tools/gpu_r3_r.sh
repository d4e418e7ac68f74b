#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "conv0 or matches_oracle" > gpurun_out/r3_r_test.log 2>&1
rc=$?
tail -5 gpurun_out/r3_r_test.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
timeout -k 10 300 python tools/bench_kernels.py conv0 2>&1 | tail -3
timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants > gpurun_out/r3_r_bench.json 2> gpurun_out/r3_r_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_r_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','ms_per_step_median']}, d['roofline']['frac'], d['roofline']['all_gemm_nt_tflops'], d['roofline'].get('measured_gemm_peak_tflops'))
PY
