#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/full_tests.log 2>&1
tail -15 gpurun_out/full_tests.log
