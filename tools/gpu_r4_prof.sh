#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command -> gpurun_out/r4_prof/ (+ per-shape medians)
mkdir -p gpurun_out/r4_prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r4_prof -o prof --output-format csv -- python3 bench.py --no-cpu-baseline --no-gemm-peak --steps 20 --warmup 5 > gpurun_out/r4_bench_under_rocprof.json 2> gpurun_out/r4_prof.err
echo rc=$?
python tools/trace_shapes.py gpurun_out/r4_prof/prof_kernel_trace.csv 48 > gpurun_out/r4_gemm_shapes.txt
head -50 gpurun_out/r4_gemm_shapes.txt
