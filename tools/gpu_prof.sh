#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command -> gpurun_out/r3_prof/ (+ the bench line it printed)
mkdir -p gpurun_out/r3_prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_prof -o prof --output-format csv -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3_bench_under_rocprof.json 2> gpurun_out/r3_prof.err
echo rc=$?
ls -R gpurun_out/r3_prof | head -20
tail -c 600 gpurun_out/r3_bench_under_rocprof.json
