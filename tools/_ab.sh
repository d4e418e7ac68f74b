set -e
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_a_dist_gpu.py -q -x 2>&1 | tail -2
for v in 0 auto 1; do
echo "PACK=$v: $(W2VS_PACK_WGRADS=$v timeout -k 10 200 python bench.py --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(d["ms_per_step"], d["ms_per_step_median"], r["achieved"], r["avg_launch_us"], r["flops_per_launch_avg"], r["launches_in_timed_region"])')"
done
