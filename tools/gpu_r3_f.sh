#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_config5_gpu.py "tests/test_model_gpu.py::test_large_full_length_step_matches_oracle" tests/test_rnnt_gpu.py -q > gpurun_out/r3_f_tests.log 2>&1
tail -30 gpurun_out/r3_f_tests.log
for f in gpurun_out/parity_config5_*.json gpurun_out/parity_large_full_length.json; do echo $f; cat $f | head -60; done
