#!/bin/bash
# Same-box A/B in the step, base and large configuration: $1 (previous build) against $2, three interleaved pairs each.
#   bash tools/ab_bench_both.sh wav2vec-s_amd/libw2vs_prev.so wav2vec-s_amd/libw2vs.so
set -e
A=${1:-wav2vec-s_amd/libw2vs_prev.so}; B=${2:-wav2vec-s_amd/libw2vs.so}
mkdir -p gpurun_out
bash tools/ab_bench.sh $A $B 3 | tee gpurun_out/ab_base.txt
bash tools/ab_bench.sh $A $B 3 --workload large | tee gpurun_out/ab_large.txt
