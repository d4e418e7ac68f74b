#!/bin/bash
# needs the tuning build (make -C wav2vec-s_amd/csrc tuning): libw2vs.so itself reads no environment variable
export W2VS_LIB=${W2VS_LIB:-$PWD/wav2vec-s_amd/libw2vs_tuning.so}
# The same in-step A/B for the LARGE configuration (24 L, d 1024, ffn 4096, pre-LN: input gradients without the residual add).
run() { W2VS_NT_FORCE="$1" timeout -k 10 200 python bench.py --workload large --steps 12 --warmup 4 --no-cpu-baseline --no-variants --no-gemm-peak 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('%-28s' % '$1', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['all_gemm_nt_tflops'])"; }
run ""
for f in 3072:1024:1=8:256 3072:1024:1=8:320 3072:1024:1=5:256 3072:1024:1=5:1160 4096:1024:7=8:256 4096:1024:7=8:320 4096:1024:7=5:1160 \
         4096:1024:8=8:256 4096:1024:8=8:320 4096:1024:8=5:1160 1024:4096:1=5:256 1024:4096:1=5:192 1024:4096:1=8:256 \
         1024:4096:0=5:256 1024:4096:0=5:192 1024:4096:0=8:256 1024:3072:0=5:256 1024:3072:0=5:192 1024:3072:0=8:256 \
         1024:1024:1=5:256 1024:1024:1=5:192 1024:1024:1=3:160 1024:1024:0=5:256 1024:1024:0=5:192 1024:1024:0=3:160; do run $f; done
run ""
