#!/bin/bash
# conv layer 0 backward, same box, three interleaved rounds (tuning build): W2VS_CONV0_BWD_PIPE = 1 the software-pipelined
# pass 1 (the product's choice), 0 = one tile pair at a time; W2VS_CONV0_BWD_MAP = 0 = the forward's round-5 channel layout
# (1, rounds 1-4 layout, is the product's choice).  The split forms of round 5 are kept as a diff:
# tools/probes/conv0_bwd_split_round5.diff
T=$PWD/wav2vec-s_amd/libw2vs_tuning.so
python -m pytest tests/test_kernels_gpu.py -q -x -k "conv0" 2>&1 | tail -1
for r in 1 2 3; do for f in 1 0; do echo -n "W2VS_CONV0_BWD_PIPE=$f "; W2VS_LIB=$T W2VS_CONV0_BWD_PIPE=$f python tools/bench_kernels.py "conv0 bwd" 2>&1 | grep "conv0 "; done; done
