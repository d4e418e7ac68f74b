#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4_phase
timeout -k 10 300 rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace -d gpurun_out/r4_phase -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-gemm-peak > gpurun_out/r4_phase/bench.json 2> gpurun_out/r4_phase/err.log
ls gpurun_out/r4_phase | head; head -2 gpurun_out/r4_phase/p_marker_api_trace.csv; head -2 gpurun_out/r4_phase/p_hip_api_trace.csv
python tools/phase_cut.py gpurun_out/r4_phase/p > gpurun_out/r4_phase_ranges.txt 2>&1; cat gpurun_out/r4_phase_ranges.txt
rm -f gpurun_out/r4_phase/p_hip_api_trace.csv gpurun_out/r4_phase/p_kernel_trace.csv
for w in fp32 bf16; do W2VS_FORCE_DIST=1 timeout -k 10 200 python bench.py --wire $w --no-cpu-baseline --no-variants --no-gemm-peak 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('force_dist wire=$w', d['ms_per_step'], d['ms_per_step_median'])"; done
timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --no-gemm-peak 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('no exchange', d['ms_per_step'], d['ms_per_step_median'])"
