#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command -> gpurun_out/prof/ (+ the bench line it printed)
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o prof --output-format csv -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/bench_under_rocprof.json 2> gpurun_out/prof.err
echo rc=$?
ls -R gpurun_out/prof | head -20
tail -c 600 gpurun_out/bench_under_rocprof.json
