set -e
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_a_dist_gpu.py -q -x 2>&1 | tail -2
for i in 1 2 3; do
for v in 0 1; do
echo "LN_DEFER=$v: $(W2VS_LN_DEFER=$v timeout -k 10 200 python bench.py --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["ms_per_step_median"])')"
done; done
