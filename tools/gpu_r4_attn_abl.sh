#!/bin/bash
# forward-attention timing-only ablations (tools/attn_probe.py one; W2VS_LIB picks the build)
mkdir -p gpurun_out
{
echo "baseline"; python tools/attn_probe.py one
for n in 1 2 4 5 8; do echo "ABL=$n"; W2VS_LIB=$PWD/wav2vec-s_amd/libw2vs_abl$n.so python tools/attn_probe.py one; done
echo "NW=4"; W2VS_ATTN_NW=4 python tools/attn_probe.py one
} > gpurun_out/r4_attn_abl.txt 2>&1
cat gpurun_out/r4_attn_abl.txt | grep -v amdgpu.ids
