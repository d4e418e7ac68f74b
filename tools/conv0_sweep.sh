#!/bin/bash
# conv layer 0 backward, same box, three interleaved rounds: the channel layouts of the kernel (tuning build: W2VS_CONV0_BWD_MAP
# = 1 rounds 1-4 layout, the product's choice; 0 = the forward's round-5 layout).  The split forms of round 5 are kept as a diff:
# tools/probes/conv0_bwd_split_round5.diff
T=$PWD/wav2vec-s_amd/libw2vs_tuning.so
for r in 1 2 3; do for f in 1 0; do echo -n "W2VS_CONV0_BWD_MAP=$f "; W2VS_LIB=$T W2VS_CONV0_BWD_MAP=$f python tools/bench_kernels.py "conv0 bwd" 2>&1 | grep "conv0 "; done; done
