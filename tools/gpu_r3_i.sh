#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3_i_bench.json 2> gpurun_out/r3_i_bench.err || { tail -5 gpurun_out/r3_i_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_i_bench.json")); r=d["roofline"]
print(d["ms_per_step"], d["ms_per_step_median"], d["value"], r["step"], d["variants"], r.get("measured_gemm_peak_tflops"))
PY
W2VS_NT8=0 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3_i_bench_nt8off.json 2> gpurun_out/r3_i_bench2.err || exit 1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_i_bench_nt8off.json")); r=d["roofline"]
print("NT8 off:", d["ms_per_step"], d["ms_per_step_median"], d["value"], d["variants"])
PY
