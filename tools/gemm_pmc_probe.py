#!/usr/bin/env python3
"""A few launches of the main GEMM kernels at the encoder shapes, for rocprofv3 --pmc passes (tools/gemm_pmc.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import ops  # noqa: E402

BF = torch.bfloat16
R, E, F = 6544, 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda r, c, s=1.0: (torch.randn(r, c, device="cuda", generator=g) * s).to(BF)      # noqa: E731
x, h = mk(R, E), mk(R, F)
wqkv, w1, w2 = mk(3 * E, E, 0.03), mk(F, E, 0.03), mk(E, F, 0.03)
b3, b1, b2 = mk(1, 3 * E).view(-1), mk(1, F).view(-1), mk(1, E).view(-1)
N = 12
for _ in range(N):
    ops.linear_fwd(x, wqkv, b3)                                  # nt8 256x256
for _ in range(N):
    ops.linear_fwd(x, w1, b1, gelu=True, save_pre=True, save_grad=True)   # nt8 320x256 + gelu
for _ in range(N):
    ops.linear_fwd(h, w2, b2)                                    # loader/consumer 160x128
dys = [mk(R, 3 * E, 0.5), mk(R, E, 0.5), mk(R, F, 0.5), mk(R, E, 0.5)]
xs = [mk(R, E, 0.5), mk(R, E, 0.5), mk(R, E, 0.5), mk(R, F, 0.5)]
ws = [torch.zeros(dy.shape[1], xx.shape[1], device="cuda") for dy, xx in zip(dys, xs)]
bs = [torch.zeros(dy.shape[1], device="cuda") for dy in dys]
prob = [dict(a=dy, b=xx, out_f32=w, M=dy.shape[1], N=xx.shape[1], K=R, lda=dy.shape[1], ldb=xx.shape[1], ldc=xx.shape[1],
             alpha=1.0, colsum_out=b) for dy, xx, w, b in zip(dys, xs, ws, bs)]
for _ in range(N):
    ops.gemm_tn_group(prob)
torch.cuda.synchronize()
print("done")
