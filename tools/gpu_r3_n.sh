#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_stream_gpu.py tests/test_joiner_gpu.py tests/test_config5_gpu.py tests/test_api_gpu.py -q -x > gpurun_out/r3_n_tests.log 2>&1; tail -4 gpurun_out/r3_n_tests.log
timeout -k 10 300 python bench.py --workload caat 2>/dev/null | tail -c 200
timeout -k 10 300 python bench.py --workload stream 2>/dev/null | tail -c 420
