#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do
  W2VS_CHECK_FINITE=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 > gpurun_out/r3_g_off$i.json 2> gpurun_out/r3_g_off$i.err || exit 1
  W2VS_CHECK_FINITE=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 > gpurun_out/r3_g_on$i.json 2> gpurun_out/r3_g_on$i.err || exit 1
done
python - <<'PY'
import json
for n in ("off1","on1","off2","on2"):
    d=json.load(open("gpurun_out/r3_g_%s.json"%n)); r=d["roofline"]
    print(n, d["ms_per_step"], d["ms_per_step_median"], d["value"], r["kernel"], r["achieved"], r.get("all_gemm_nt_tflops"), r.get("all_gemm_tn_tflops"), r["step"]["frac"], r.get("measured_gemm_peak_tflops"))
PY
timeout -k 10 300 python bench.py --workload large --length-mix --no-cpu-baseline --no-variants --steps 10 --warmup 3 > gpurun_out/r3_g_large_mix.json 2> gpurun_out/r3_g_large_mix.err || { tail -5 gpurun_out/r3_g_large_mix.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3_g_large_mix.json')); print(d['ms_per_step'], d['ms_per_step_median'], d['value'], d['config']['length_mix'])"
