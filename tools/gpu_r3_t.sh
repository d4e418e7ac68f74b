#!/bin/bash
# sanity sweep of every bench workload + the forced-distributed path + smoke()
mkdir -p gpurun_out
set -o pipefail
W2VS_FORCE_DIST=1 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-variants --no-gemm-peak 2>gpurun_out/t_dist.err | tail -c 300 || { echo "forced dist failed"; tail -5 gpurun_out/t_dist.err; exit 1; }
echo
for w in large stream data rnnt caat; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline 2>gpurun_out/t_$w.err | tail -c 260 || { echo "$w failed"; tail -5 gpurun_out/t_$w.err; exit 1; }
  echo
done
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
