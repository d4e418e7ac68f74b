#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_model_gpu.py tests/test_a_dist_gpu.py tests/test_api_gpu.py -q -x > gpurun_out/r3_p_test.log 2>&1
rc=$?
tail -15 gpurun_out/r3_p_test.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r3_p_bench.json 2> gpurun_out/r3_p_bench.err
tail -c 1500 gpurun_out/r3_p_bench.json
