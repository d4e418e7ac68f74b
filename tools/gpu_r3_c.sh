#!/bin/bash
# round 3, step C: trainer / stream / joiner / rnnt / kernel tests after the ADVICE fixes; records per-family errors
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_a_dist_gpu.py tests/test_stream_gpu.py tests/test_joiner_gpu.py tests/test_rnnt_gpu.py "tests/test_kernels_gpu.py::test_group_attention_beyond_one_record_table" -q > gpurun_out/r3_c_tests.log 2>&1
tail -25 gpurun_out/r3_c_tests.log
for f in gpurun_out/parity_stream_*.json gpurun_out/parity_joiner_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['by_family'], d['median'])"; done
