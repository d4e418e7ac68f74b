#!/usr/bin/env python3
"""v1 vs v2 attention outputs / gradients (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import wav2vec_s_amd  # noqa
from wav2vec_s_amd import ops
BF = torch.bfloat16
for (Tp, m, r, H, p) in [(48, 16, 8, 2, 0.0), (546, 16, 8, 2, 0.0), (546, 16, 8, 2, 0.1)]:
    B = 2
    N = Tp + (Tp // m) * r
    g = torch.Generator().manual_seed(Tp)
    qkv = torch.randn(B, N, 3 * H * 64, generator=g).to(BF).cuda()
    dout = torch.randn(B, N, H * 64, generator=g).to(BF).cuda()
    res = {}
    for v in (1, 2):
        ops.attn_tune(v)
        o, lse = ops.attn_fwd(qkv, H, Tp, m, r, p_drop=p, seed=3)
        d = ops.attn_bwd(dout, qkv, o, lse, H, Tp, m, r, p_drop=p, seed=3)
        res[v] = (o.float().cpu(), d.float().cpu())
    ops.attn_tune(-1)
    C = H * 64
    for nm, sl in (("dq", slice(0, C)), ("dk", slice(C, 2 * C)), ("dv", slice(2 * C, 3 * C))):
        a, b = res[1][1][..., sl], res[2][1][..., sl]
        e = (a - b).view(B, N, H, 64).abs().amax(-1)
        bad = (e > 0.05 * a.abs().max()).nonzero()
        print((Tp, m, r, H, p), nm, "rel", float((a - b).norm() / a.norm()), "bad rows", bad.shape[0], bad[:6].tolist())

print("---- stored keep masks vs re-hash")
for (Tp, m, r, H) in [(130, 32, 16, 2), (546, 16, 8, 3)]:
    B = 2
    N = Tp + (Tp // m) * r
    g = torch.Generator().manual_seed(Tp)
    qkv = torch.randn(B, N, 3 * H * 64, generator=g).to(BF).cuda()
    dout = torch.randn(B, N, H * 64, generator=g).to(BF).cuda()
    o1, l1 = ops.attn_fwd(qkv, H, Tp, m, r, p_drop=0.2, seed=77)
    g1 = ops.attn_bwd(dout, qkv, o1, l1, H, Tp, m, r, p_drop=0.2, seed=77).float().cpu()
    bits = ops.attn_drop_bits(B, H, N)
    bits.fill_(0x5A5A5A5A)
    o2, l2 = ops.attn_fwd(qkv, H, Tp, m, r, p_drop=0.2, seed=77, drop_bits=bits)
    g2 = ops.attn_bwd(dout, qkv, o2, l2, H, Tp, m, r, p_drop=0.2, seed=77, drop_bits=bits).float().cpu()
    print("o equal", torch.equal(o1, o2), "lse equal", torch.equal(l1, l2))
    C = H * 64
    for nm, sl in (("dq", slice(0, C)), ("dk", slice(C, 2 * C)), ("dv", slice(2 * C, 3 * C))):
        a, b = g1[..., sl], g2[..., sl]
        e = (a - b).view(B, N, H, 64).abs().amax(-1)
        bad = (e > 0.02 * a.abs().max()).nonzero()
        print((Tp, m, r, H), nm, "rel", float((a - b).norm() / a.norm()), "maxabs", float((a - b).abs().max()), "bad rows", bad.shape[0], bad[:8].tolist())
