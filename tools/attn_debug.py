#!/usr/bin/env python3
"""v1 vs v2 attention outputs, row by row (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import wav2vec_s_amd  # noqa
from wav2vec_s_amd import ops
import w2vs_oracle as O
BF = torch.bfloat16
for (Tp, m, r, H) in [(50, 8, 4, 3), (300, 24, 6, 2), (48, 16, 8, 2)]:
    B = 2
    N = Tp + (Tp // m) * r
    g = torch.Generator().manual_seed(Tp)
    qkv = torch.randn(B, N, 3 * H * 64, generator=g).to(BF).cuda()
    rc_idx, rc_oob, _ = O.block_structure(Tp, m, r)
    pad = torch.zeros(B, Tp, dtype=torch.bool); pad[1, Tp - 1] = True
    kpad = torch.cat([pad, pad.index_select(1, rc_idx) | rc_oob.unsqueeze(0)], dim=1) if r > 0 else pad
    kp = kpad.to(torch.uint8).cuda()
    outs = {}
    for v in (1, 2):
        ops.attn_tune(v)
        o, lse = ops.attn_fwd(qkv, H, Tp, m, r, kpad=kp)
        outs[v] = (o.float().cpu(), lse.cpu())
    ops.attn_tune(-1)
    d = (outs[1][0] - outs[2][0]).view(B, N, H, 64).abs().amax(-1)      # [B, N, H]
    bad = (d > 0.02).nonzero()
    print("shape", (Tp, m, r, H), "N", N, "bad rows:", bad.shape[0])
    for row in bad[:40].tolist():
        b, q, h = row
        print("   b%d q%d h%d  maxdiff %.3f  lse v1 %.4f v2 %.4f  qpad=%d" % (b, q, h, float(d[b, q, h]), float(outs[1][1][b, h, q]),
                                                                   float(outs[2][1][b, h, q]), int(kpad[b, q])))
