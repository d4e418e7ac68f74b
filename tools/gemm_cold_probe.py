#!/usr/bin/env python3
"""NT GEMMs of the encoder layer, timed WARM (the same operands launched back to back, what tools/gemm_probe.py reports) and
COLD (a 1 GiB write to another buffer between launches, every launch bracketed by its own events): in a training step
a GEMM's operands were last touched many kernels ago.   python tools/gemm_cold_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import ops  # noqa: E402

BF = torch.bfloat16
R = 6544
g = torch.Generator(device="cuda").manual_seed(0)
flush = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")


def timed(fn, cold, iters=12):
    ts = []
    for i in range(iters + 2):
        if cold:
            flush.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for name, N, K in (("qkv fwd", 2304, 768), ("fc1 fwd", 3072, 768), ("fc2 fwd", 768, 3072), ("out fwd", 768, 768)):
    x = torch.randn(R, K, device="cuda", generator=g).to(BF)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(BF)
    b = torch.randn(N, device="cuda", generator=g).to(BF)
    for cfg_name, tune in (("default", (-1, 0)), ("8ph 256", (8, 256)), ("8ph 320", (8, 320)), ("persist 160x256", (5, 1160)), ("persist 256x128", (5, 256)),
                           ("persist 160x128", (5, 160)), ("lc 160x128", (3, 160))):
        if tune[1] == 1160 and N % 256:
            continue
        ops.gemm_tune(*tune)
        tw = timed(lambda: ops.linear_fwd(x, w, b), False)
        tc = timed(lambda: ops.linear_fwd(x, w, b), True)
        fl = 2.0 * R * N * K
        print("%-8s %-16s warm %6.1f us (%5.0f TF/s)   cold %6.1f us (%5.0f TF/s)" % (name, cfg_name, tw, fl / tw / 1e6, tc, fl / tc / 1e6), flush=True)
    ops.gemm_tune(-1, 0)
