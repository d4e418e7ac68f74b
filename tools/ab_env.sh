#!/bin/bash
# Same-box A/B of a Python-level switch in the step: bench.py alternately with $1=0 and $1=1, $2 rounds (default 3), extra flags behind.
#   bash tools/ab_env.sh W2VS_OVERWRITE_WGRADS 3 [--workload large]
V=${1:?env var}; N=${2:-3}; shift 2
for i in $(seq $N); do
  for x in 0 1; do
    env $V=$x timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --no-gemm-peak "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$V=$x  mean %.3f median %.3f ms  min/max %s' % (d['ms_per_step'], d['ms_per_step_median'], d.get('ms_per_step_min_max')))"
  done
done
