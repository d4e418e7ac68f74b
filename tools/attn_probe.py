#!/usr/bin/env python3
"""Attention kernel probe: times w2vs_attn_fwd / w2vs_attn_bwd at the cfgB shape under ablations that separate the costs
(dropout hash, block-causal imbalance, masking).  python tools/attn_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import flops, ops  # noqa: E402

BF = torch.bfloat16
dev = "cuda"


def t_us(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(tag, B, H, Tp, m, r, p):
    N = Tp + (Tp // m) * r
    E = 64 * H
    qkv = (torch.randn(B, N, 3 * E, device=dev)).to(BF)
    dout = torch.randn(B, N, E, device=dev).to(BF)
    keep = os.environ.get("W2VS_ATTN_KEEP_BITS", "0") == "1"         # as the engine: the keep-mask store is off by default
    bits = ops.attn_drop_bits(B, H, N) if (p > 0 and keep) else None
    o, lse = ops.attn_fwd(qkv, H, Tp, m, r, p_drop=p, seed=5, drop_bits=bits)
    pairs = flops.attention_pairs(Tp, m, r)
    f = 4.0 * pairs * 64 * H * B
    tf = t_us(lambda: ops.attn_fwd(qkv, H, Tp, m, r, p_drop=p, seed=5, drop_bits=bits))
    tb = t_us(lambda: ops.attn_bwd(dout, qkv, o, lse, H, Tp, m, r, p_drop=p, seed=5, drop_bits=bits))
    print("%-34s N=%4d pairs=%7d  fwd %6.1f us (%6.1f TF/s)   bwd(dq+dkv) %6.1f us (%6.1f TF/s)" % (
        tag, N, pairs, tf, f / tf / 1e6, tb, 2.5 * f / tb / 1e6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":      # for rocprofv3 --pmc runs: one shape, few launches
        run("cfgB m16 r8 p0.1", 8, 12, 546, 16, 8, 0.1)
        sys.exit(0)
    run("cfgB m16 r8 p0.1", 8, 12, 546, 16, 8, 0.1)
    run("cfgB m16 r8 p0", 8, 12, 546, 16, 8, 0.0)
    run("dense N=818 (m=818,r=0) p0.1", 8, 12, 818, 818, 0, 0.1)
    run("dense N=818 (m=818,r=0) p0", 8, 12, 818, 818, 0, 0.0)
    run("cfgB m32 r16 p0.1", 8, 12, 546, 32, 16, 0.1)
    run("large 3x999 m16 r8 p0.1 H16", 3, 16, 1000, 16, 8, 0.1)
