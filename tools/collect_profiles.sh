#!/bin/bash
# Round-end evidence on the GPU box: kernel stats of the headline bench + the HBM-traffic counter passes the bench line
# quotes (separate --pmc runs, kernel trace only - the pool refuses --pmc together with the other trace domains).
#   gpurun -- 'bash tools/collect_profiles.sh'   then copy gpurun_out/final_* into profiles/roundN_*
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/final_prof -o prof --output-format csv -- python3 bench.py --no-cpu-baseline --no-gemm-peak > gpurun_out/final_bench_under_rocprof.json 2> gpurun_out/final_prof.err || exit 1
cp gpurun_out/final_prof/prof_kernel_stats.csv gpurun_out/final_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/final_pmc_f -o p --output-format rocpd -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-gemm-peak > gpurun_out/final_pmc_f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/final_pmc_w -o p --output-format rocpd -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-gemm-peak > gpurun_out/final_pmc_w.log 2>&1 || exit 1
python tools/pmc_traffic.py gpurun_out/final_pmc_f/p_results.db gpurun_out/final_pmc_w/p_results.db > gpurun_out/final_pmc_hbm_traffic.json || exit 1
cp gpurun_out/final_pmc_hbm_traffic.json profiles/round${ROUND:-3}_pmc_hbm_traffic.json   # so that the bench line below quotes it (same gemm.hip)
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || exit 1
head -c 600 gpurun_out/final_bench.json
