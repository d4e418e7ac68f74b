#!/bin/bash
# Round-end evidence on the GPU box: kernel stats of the headline bench + the HBM-traffic counter passes the bench line
# quotes (separate --pmc runs, kernel trace only - the pool refuses --pmc together with the other trace domains).
#   gpurun -- 'bash tools/collect_profiles.sh'   then copy gpurun_out/final_* into profiles/roundN_*
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/final_prof -o prof --output-format csv -- python3 bench.py --no-cpu-baseline --no-gemm-peak > gpurun_out/final_bench_under_rocprof.json 2> gpurun_out/final_prof.err || exit 1
cp gpurun_out/final_prof/prof_kernel_stats.csv gpurun_out/final_kernel_stats.csv
python tools/trace_shapes.py gpurun_out/final_prof/prof_kernel_trace.csv 48 > gpurun_out/final_gemm_shapes.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/final_pmc_f -o p --output-format rocpd -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-gemm-peak > gpurun_out/final_pmc_f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/final_pmc_w -o p --output-format rocpd -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-gemm-peak > gpurun_out/final_pmc_w.log 2>&1 || exit 1
python tools/pmc_traffic.py gpurun_out/final_pmc_f/p_results.db gpurun_out/final_pmc_w/p_results.db > gpurun_out/final_pmc_hbm_traffic.json || exit 1
cp gpurun_out/final_pmc_hbm_traffic.json profiles/round${ROUND:-5}_pmc_hbm_traffic.json   # so that the bench line below quotes it (same gemm.hip)
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || exit 1
head -c 600 gpurun_out/final_bench.json

# tool-level evidence quoted in DESIGN.md: tile probe, attention probe, grouped-wgrad probe with its ablations
python tools/gemm_probe.py fc1 qkv fc2 out conv1 conv2 sq4k sq8k > gpurun_out/final_gemm_probe.txt 2>/dev/null
python tools/attn_probe.py > gpurun_out/final_attention_probe.txt 2>/dev/null
# selector switches and timing-only ablations live in the tuning build only (make -C wav2vec-s_amd/csrc tuning; travels with the snapshot)
if [ -f wav2vec-s_amd/libw2vs_tuning.so ]; then
( export W2VS_LIB=$PWD/wav2vec-s_amd/libw2vs_tuning.so
  for r in 6544 5584 4800; do W2VS_TN8=0 python tools/wgrad_group_probe.py $r; W2VS_TN8=1 python tools/wgrad_group_probe.py $r; done
  for d in 1 2 3; do W2VS_TN8=1 W2VS_TN8_DBG=$d python tools/wgrad_group_probe.py 6544; done
  W2VS_TN8=1 W2VS_TN8_S=1 python tools/wgrad_group_probe.py 6544 ) > gpurun_out/final_wgrad_group_probe.txt 2>/dev/null
fi
tail -3 gpurun_out/final_wgrad_group_probe.txt

# the rows around the headline path: large configuration, config-5 step, streaming encoder call
python bench.py --workload large --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final_bench_large.json
python bench.py --workload caat 2>/dev/null | tail -1 > gpurun_out/final_bench_caat.json
python bench.py --workload stream 2>/dev/null | tail -1 > gpurun_out/final_bench_stream.json
tail -c 200 gpurun_out/final_bench_caat.json
