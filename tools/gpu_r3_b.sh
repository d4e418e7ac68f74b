#!/bin/bash
# bench A/B of a feature switch given as $1 (env var name), then kernel + model + trainer tests
mkdir -p gpurun_out
V=${1:-W2VS_TN8}
for i in 1 2; do
  env $V=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 > gpurun_out/r3_b_off$i.json 2> gpurun_out/r3_b_off$i.err || exit 1
  env $V=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 > gpurun_out/r3_b_on$i.json 2> gpurun_out/r3_b_on$i.err || exit 1
done
python - <<'PY'
import json
for n in ("off1","on1","off2","on2"):
    d=json.load(open("gpurun_out/r3_b_%s.json"%n)); r=d["roofline"]
    print(n, d["ms_per_step"], d["value"], r["kernel"], r["achieved"], r.get("all_gemm_nt_tflops"), r.get("all_gemm_tn_tflops"), r["step"]["frac"])
PY
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_a_dist_gpu.py -x -q > gpurun_out/r3_b_tests.log 2>&1
tail -5 gpurun_out/r3_b_tests.log
