#!/usr/bin/env python3
"""A/B of NT GEMM tile configurations at the encoder shapes (cfgB: R = 6544 rows).  python tools/gemm_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import ops  # noqa: E402

BF = torch.bfloat16


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


R = 6544
g = torch.Generator(device="cuda").manual_seed(0)
SHAPES = (("fc1 fwd", 3072, 768), ("qkv fwd", 2304, 768), ("fc2 fwd", 768, 3072), ("out fwd", 768, 768),
          ("conv1 fwd", 512, 1536, 140000), ("conv1 dgrad", 1024, 1024, 140000), ("conv2 fwd", 512, 1536, 70000),
          ("sq4k fwd", 4096, 4096, 4096), ("sq8k fwd", 8192, 8192, 8192),
          ("last fc2", 768, 3072, 2080), ("last fc1", 3072, 768, 2080), ("last out", 768, 768, 2080), ("conv4 fwd", 512, 1536, 17500), ("conv5 fwd", 512, 1024, 8744))
if len(sys.argv) > 1:
    SHAPES = tuple(s for s in SHAPES if s[0].split()[0] in sys.argv[1:])
for name, N, K, *rest in SHAPES:
    R = rest[0] if rest else 6544
    x = torch.randn(R, K, device="cuda", generator=g).to(BF)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(BF)
    b = torch.randn(N, device="cuda", generator=g).to(BF)
    aux = torch.randn(R, N, device="cuda", generator=g).to(BF)
    ref = (x.float() @ w.float().t() + b.float())
    for cfg_name, tune in (("default", (-1, 0)), ("8ph 256x256", (8, 256)), ("8ph 320x256", (8, 320)), ("128^2 dma", (2, 0)), ("lc 256x128", (3, 256)), ("lc 160x128", (3, 160)), ("lc 160x256", (3, 1160)), ("persist 256x128", (5, 256)), ("persist 192x128", (5, 192)), ("lc 192x128", (3, 192)), ("128^2 regs", (0, 0)), ("128^2 mode1", (1, 0)), ("persist 160x128", (5, 160)),
                           ("persist 160x256", (5, 1160)), ("persist 160x256 / 4 consumer waves (tuning build)", (5, 2160))):
        if tune[1] in (1160, 2160) and N % 256:
            continue
        if tune[1] == 2160 and "tuning" not in os.environ.get("W2VS_LIB", ""):
            continue
        ops.gemm_tune(*tune)
        y = ops.linear_fwd(x, w, b)
        err = float((y.float() - ref).norm() / ref.norm())
        t1 = t_us(lambda: ops.linear_fwd(x, w, b))
        t2 = t_us(lambda: ops.linear_fwd(x, w, b, gelu=True, save_pre=True, save_grad=True))
        t3 = t_us(lambda: ops.linear_dgrad(x, w, mul_aux=aux))          # same shape class: [R,K] x [N,K]^T with EPI_MUL
        t4 = t_us(lambda: ops.linear_dgrad(x, w, add_aux=aux))          # ... with EPI_ADD (the residual-add dgrads)
        fl = 2.0 * R * N * K
        print("%-8s %-11s rel err %.1e   bias %6.1f us (%5.0f TF/s)   gelu+saveg %6.1f us   mul-aux %6.1f us   add-aux %6.1f us" % (
            name, cfg_name, err, t1, fl / t1 / 1e6, t2, t3, t4), flush=True)
    ops.gemm_tune(-1, 0)
