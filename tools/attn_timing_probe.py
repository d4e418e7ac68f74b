#!/usr/bin/env python3
"""Reads the per-workgroup phase timings an INSTRUMENTED build of attn2_fwd_kernel leaves in the lse buffer (s_memtime at
the kernel's entry, loop entry, after the first sub-tile, loop exit, merge barrier, end - see DESIGN.md 5, attention).
Only meaningful with such a build passed through W2VS_LIB; the shipped library writes the real lse."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import ops  # noqa: E402

B, H, Tp, m, r = 8, 12, 546, 16, 8
N = Tp + (Tp // m) * r
for p in (0.1, 0.0):
    qkv = torch.randn(B, N, 3 * 64 * H, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        o, lse = ops.attn_fwd(qkv, H, Tp, m, r, p_drop=p, seed=5)
    torch.cuda.synchronize()
    L = lse.float().cpu().numpy().reshape(B * H, -1)          # [BH, Ns]
    nqt = (N + 31) // 32
    rec = np.stack([L[:, 32 * t: 32 * t + 11] for t in range(nqt - 1)], 1).reshape(-1, 11)   # skip the ragged last tile
    pro, first, loop, bar, merge, nT, t0, t4, bid, xcc, cu = rec.T
    print("p=%.1f  WGs %d" % (p, len(rec)))
    for name, v in (("prologue", pro), ("first sub-tile", first), ("loop", loop), ("barrier wait", bar), ("merge+store", merge)):
        print("  %-16s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f ticks" % (
            name, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
    per = loop / np.maximum(np.ceil(nT / 4), 1)
    print("  loop per wave-0 sub-tile: mean %.0f p50 %.0f ticks; nT mean %.1f max %d" % (per.mean(), np.median(per), nT.mean(), nT.max()))
    life = (t4 - t0) % (1 << 24)
    print("  WG lifetime mean %.0f p50 %.0f max %.0f ticks" % (life.mean(), np.median(life), life.max()))
