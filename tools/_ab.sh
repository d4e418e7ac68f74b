set -e
for i in 1 2; do
for v in 0 1 2 4 8 16 5 31; do
echo "NTX=$v: $(W2VS_NTX=$v timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["ms_per_step_median"])')"
done; done
