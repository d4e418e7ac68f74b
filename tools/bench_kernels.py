#!/usr/bin/env python3
"""Micro-benchmarks of the individual libw2vs kernels at the cfgB shapes (8 x 175000 samples:
T=546, N=818, R=6544 token rows).  python tools/bench_kernels.py [filter]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = "cuda"
flt = sys.argv[1] if len(sys.argv) > 1 else ""


def timeit(name, fn, flops=None, bytes_=None, iters=20):
    if flt and flt not in name:
        return
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    extra = ""
    if flops:
        extra += "  %7.1f TFLOP/s" % (flops / us / 1e6)
    if bytes_:
        extra += "  %7.1f GB/s" % (bytes_ / us / 1e3)
    print("%-44s %9.1f us%s" % (name, us, extra), flush=True)


def r(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(BF)


B, T, Tp, m_, r_ = 8, 546, 546, 16, 8
N = Tp + (Tp // m_) * r_
R, E, F, H = B * N, 768, 3072, 12

# ---- GEMMs
for (nm, M, Nn, K) in [("qkv", R, 3 * E, E), ("out", R, E, E), ("fc1", R, F, E), ("fc2", R, E, F)]:
    x, w, b = r(M, K), r(Nn, K, scale=0.02), r(Nn)
    timeit("gemm_nt fwd %s %dx%dx%d" % (nm, M, Nn, K), lambda: ops.linear_fwd(x, w, b), flops=2.0 * M * Nn * K)
    if nm == "fc1":
        timeit("gemm_nt fwd %s gelu_save" % nm, lambda: ops.linear_fwd(x, w, b, gelu=True, save_pre=True), flops=2.0 * M * Nn * K)
    dy = r(M, Nn)
    dw = torch.zeros(Nn, K, device=dev)
    timeit("gemm_tn wgrad %s" % nm, lambda: ops.linear_wgrad(dy, x, dw), flops=2.0 * M * Nn * K)
    timeit("colsum %s" % nm, lambda: ops.colsum(dy, torch.zeros(Nn, device=dev)), bytes_=M * Nn * 2)

# ---- conv stack
Ls = [34999, 17499, 8749, 4374, 2186, 1093, 546]
ks = [(3, 2)] * 4 + [(2, 2)] * 2
for i, (k, s) in enumerate(ks):
    Lin, Lout = Ls[i], Ls[i + 1]
    x = r(B, Lin, 512)
    w2 = r(512, k * 512, scale=0.03)
    timeit("conv%d fwd (k%d s%d) L=%d" % (i + 1, k, s, Lout), lambda: ops.conv_cl_fwd(x, w2, k, s), flops=2.0 * B * Lout * 512 * 512 * k)
    dy = r(B, Lout, 512)
    pre = r(B, Lin, 512)
    timeit("conv%d dgrad" % (i + 1), lambda: ops.conv_cl_dgrad(dy, w2, k, s, Lin, dgelu_aux=pre), flops=2.0 * B * Lout * 512 * 512 * k)
    dw = torch.zeros(512, k * 512, device=dev)
    timeit("conv%d wgrad" % (i + 1), lambda: ops.conv_cl_wgrad(dy, x, k, s, dw), flops=2.0 * B * Lout * 512 * 512 * k)

wave = r(B, 175000)
w0, g0, b0 = r(512, 1, 10, scale=0.3), r(512), r(512)
timeit("conv0 fwd", lambda: ops.conv0_fwd(wave, w0, g0, b0, 10, 5), bytes_=B * 34999 * 512 * 2)
y0, mean0, rstd0 = ops.conv0_fwd(wave, w0, g0, b0, 10, 5)
dy0 = r(B, 34999, 512)
dw0 = torch.zeros(512, 10, device=dev); dg0 = torch.zeros(512, device=dev); db0 = torch.zeros(512, device=dev)
timeit("conv0 bwd", lambda: ops.conv0_bwd(wave, w0, g0, b0, mean0, rstd0, dy0, 10, 5, dw0, dg0, db0), bytes_=B * 34999 * 512 * 2)

# ---- LN
x, res = r(R, E), r(R, E)
g, b = r(E), r(E)
timeit("ln_fwd add+drop+ln 768", lambda: ops.ln_fwd(x, g, b, res=res, want_sum=True, p_drop=0.1, seed=1), bytes_=R * E * 2 * 4)
y, s_, mean, rstd = ops.ln_fwd(x, g, b, res=res, want_sum=True, p_drop=0.1, seed=1)
dgm, dbt = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
dy = r(R, E)
timeit("ln_bwd 768", lambda: ops.ln_bwd(s_, g, b, mean, rstd, dgm, dbt, dy=dy, want_dres=True, p_drop=0.1, seed=1), bytes_=R * E * 2 * 4)

# ---- attention
qkv = r(B, N, 3 * E)
kpad = None
flops_masked = 0
for q in range(N):
    bq = q // m_ if q < Tp else (q - Tp) // r_
    flops_masked += min((bq + 1) * m_, Tp) + (r_ if Tp + (bq + 1) * r_ <= N else 0)
fl = 4.0 * flops_masked * 64 * B * H
for p in (0.0, 0.1):
    timeit("attn_fwd N=%d p=%.1f" % (N, p), lambda: ops.attn_fwd(qkv, H, Tp, m_, r_, kpad=kpad, p_drop=p, seed=3), flops=fl)
    o, lse = ops.attn_fwd(qkv, H, Tp, m_, r_, kpad=kpad, p_drop=p, seed=3)
    do = r(B, N, E)
    timeit("attn_bwd N=%d p=%.1f" % (N, p), lambda: ops.attn_bwd(do, qkv, o, lse, H, Tp, m_, r_, kpad=kpad, p_drop=p, seed=3), flops=2.5 * fl)

# ---- heads
M = 245
RM = B * M
xf, yq = r(RM, 256), r(RM, 256)
neg = torch.randint(0, M - 1, (B, 100 * M)) + (torch.arange(B) * M).unsqueeze(1)
neg = neg.to(dev)
timeit("nce_fwd", lambda: ops.nce_fwd(xf, yq, neg, B, M, 100, 0.1))
lg, nr = ops.nce_fwd(xf, yq, neg, B, M, 100, 0.1)
out3, dl = ops.ce_rows(lg)
timeit("nce_bwd", lambda: ops.nce_bwd(dl, lg, nr, xf, yq, neg, B, M, 100, 0.1))
nparam = 90325120
p32 = torch.zeros(nparam, device=dev); mm = torch.zeros_like(p32); vv = torch.zeros_like(p32); gg = torch.ones_like(p32)
p16 = torch.zeros(nparam, device=dev, dtype=BF)
timeit("adam 90M", lambda: ops.adam_step(p32, p16, mm, vv, gg, lr=1e-3, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.01, step=1),
       bytes_=nparam * (4 * 7 + 2))
