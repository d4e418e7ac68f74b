#!/usr/bin/env python3
"""Small weight-gradient GEMMs: the 128^2 atomics kernel against the loader/consumer kernel + summing launch (rocprofv3
kernel durations are the measure: run under `rocprofv3 --kernel-trace --stats`).  python tools/wgrad_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import ops  # noqa: E402

BF = torch.bfloat16


def t_us(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (N, K, R) in ((768, 512, 4368), (640, 512, 2080), (256, 768, 2080), (256, 256, 2080), (768, 768, 2080), (512, 768, 4368)):
    dy = torch.randn(R, N, device="cuda").to(BF)
    x = torch.randn(R, K, device="cuda").to(BF)
    dw = torch.zeros(N, K, device="cuda")
    db = torch.zeros(N, device="cuda")
    ref = dy.float().t() @ x.float()
    out = []
    for name, lc in (("atomics", 0), ("loader/consumer", 1), ("auto", -1)):
        ops.gemm_tune(-1, 0, lc)
        dw.zero_()
        ops.linear_wgrad(dy, x, dw, 1.0, db)
        err = float((dw - ref).norm() / ref.norm())
        out.append("%s %6.1f us (err %.0e)" % (name, t_us(lambda: ops.linear_wgrad(dy, x, dw, 1.0, db)), err))
    ops.gemm_tune(-1, 0, -1)
    print("dW[%4d,%4d] over %5d rows: %s" % (N, K, R, "   ".join(out)), flush=True)
