#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/profile_stream_call.py > gpurun_out/r3_stream_profile.log 2>&1; head -12 gpurun_out/r3_stream_profile.log
timeout -k 10 600 python -m pytest tests/test_stream_gpu.py tests/test_api_gpu.py tests/test_config5_gpu.py -q -x > gpurun_out/r3_j_tests.log 2>&1; tail -4 gpurun_out/r3_j_tests.log
timeout -k 10 300 python bench.py --workload stream > gpurun_out/r3_stream_bench.json 2> gpurun_out/r3_stream_bench.err; tail -c 900 gpurun_out/r3_stream_bench.json
