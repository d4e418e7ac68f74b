#!/usr/bin/env python3
"""Does a GEMM slow down under sustained load (DVFS)?  The QKV-shaped NT GEMM launched back to back for ~1.5 s, timed in
chunks.   python tools/sustain_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import ops  # noqa: E402

BF = torch.bfloat16
R, N, K = 6544, 2304, 768
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(R, K, device="cuda", generator=g).to(BF)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(BF)
b = torch.randn(N, device="cuda", generator=g).to(BF)
ops.linear_fwd(x, w, b)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for c in range(40):
    for _ in range(1000):
        ops.linear_fwd(x, w, b)
    ev[c + 1].record()
torch.cuda.synchronize()
ts = [ev[c].elapsed_time(ev[c + 1]) for c in range(40)]       # ms per 1000 launches = us per launch
print("us per launch, chunks of 1000:", " ".join("%.1f" % t for t in ts))
