#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -x -m gpu > gpurun_out/r3_q_test.log 2>&1
rc=$?
tail -8 gpurun_out/r3_q_test.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3_q_bench.json 2> gpurun_out/r3_q_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_q_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','ms_per_step_median']}, d['roofline']['frac'], d['roofline']['all_gemm_nt_tflops'], d['variants'])
PY
timeout -k 10 300 python bench.py --workload caat 2>/dev/null | tail -c 200
