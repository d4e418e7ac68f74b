#!/bin/bash
# needs the tuning build (make -C wav2vec-s_amd/csrc tuning): libw2vs.so itself reads no environment variable
export W2VS_LIB=${W2VS_LIB:-$PWD/wav2vec-s_amd/libw2vs_tuning.so}
# In-step A/B of NT tile choices: W2VS_NT_FORCE="N:K:epi=mode:height" per encoder GEMM class, one bench run each.
# (epi: 0 none 1 bias 6 +aux 7 gelu+gelu' 8 x aux; mode 8 = 8-phase (256|320), 5 = persistent (256|192|160|1160), 3 = one tile per WG)
run() { W2VS_NT_FORCE="$1" timeout -k 10 200 python bench.py --no-cpu-baseline --no-variants --no-gemm-peak 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('%-28s' % '$1', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['all_gemm_nt_tflops'])"; }
run ""
for f in 2304:768:1=8:320 2304:768:1=5:256 2304:768:1=5:1160 3072:768:7=8:256 3072:768:7=5:1160 3072:768:8=8:256 \
         768:3072:1=5:256 768:3072:1=5:192 768:3072:1=8:256 768:3072:6=5:256 768:3072:6=5:192 768:2304:6=5:256 768:2304:6=5:192 \
         768:768:1=5:256 768:768:1=5:192 768:768:0=5:256 768:768:0=5:192 768:768:1=3:160 768:768:0=3:160; do run $f; done
run ""
