#!/usr/bin/env python3
"""Probe ln_bwd variants under rocprofv3 (kernel durations are read from the trace in call order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wav2vec_s_amd  # noqa
from wav2vec_s_amd import ops
BF = torch.bfloat16
R, E = 6544, 768
x = torch.randn(R, E, device="cuda").to(BF); dy = torch.randn(R, E, device="cuda").to(BF)
g = torch.randn(E, device="cuda").to(BF); b = torch.randn(E, device="cuda").to(BF)
y, s_, mean, rstd = ops.ln_fwd(x, g, b, res=x, want_sum=True, p_drop=0.1, seed=1)
dg = torch.zeros(E, device="cuda"); db = torch.zeros(E, device="cuda")
variants = [dict(want_dres=True, p_drop=0.1), dict(want_dres=True, p_drop=0.0), dict(want_dres=False, p_drop=0.1),
            dict(want_dres=False, p_drop=0.0), dict(want_dres=False, want_dx=False, p_drop=0.0)]
for v in variants:
    for _ in range(10):
        ops.ln_bwd(s_, g, b, mean, rstd, dg, db, dy=dy, seed=1, **v)
    torch.cuda.synchronize()
    ops.ARENA.reset() if hasattr(ops, "ARENA") and hasattr(ops.ARENA, "reset") else None
print("variants:", variants)
