#!/usr/bin/env python3
"""The eight weight gradients of TWO encoder layers as one grouped launch (the S = 1 form of gemm_tn8_group_kernel) at the cfgB
shape.  W2VS_TN8_DBG: timing-only ablations (bit 0 no memory traffic, bit 1 no MFMAs, bit 2 a third fewer fragment reads).
python tools/wgrad_pair_probe.py [R]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import ops  # noqa: E402

BF = torch.bfloat16
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6544
E, F = 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda r, c: (torch.randn(r, c, device="cuda", generator=g) * 0.5).to(BF)      # noqa: E731
prob, chk = [], []
for layer in range(2):
    dys = [mk(R, F), mk(R, E), mk(R, 3 * E), mk(R, E)]
    xs = [mk(R, E), mk(R, F), mk(R, E), mk(R, E)]
    for dy, x in zip(dys, xs):
        w = torch.zeros(dy.shape[1], x.shape[1], device="cuda")
        b = torch.zeros(dy.shape[1], device="cuda")
        prob.append(dict(a=dy, b=x, out_f32=w, M=dy.shape[1], N=x.shape[1], K=R, lda=dy.shape[1], ldb=x.shape[1], ldc=x.shape[1],
                         alpha=1.0, colsum_out=b))
        chk.append((dy, x, w, b))
ops.gemm_tn_group(prob)
torch.cuda.synchronize()
errs = max(float((w - dy.float().t() @ x.float()).norm() / (dy.float().t() @ x.float()).norm()) for dy, x, w, b in chk)
berr = max(float((b - dy.float().sum(0)).norm() / dy.float().sum(0).norm()) for dy, x, w, b in chk)
for _ in range(5):
    ops.gemm_tn_group(prob)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40):
    ops.gemm_tn_group(prob)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 40 * 1e3
fl = sum(2.0 * R * dy.shape[1] * x.shape[1] for dy, x, w, b in chk)
print("pair, W2VS_TN8_DBG=%s R=%d: %.1f us = %.0f TF/s   worst rel err dW %.1e  db %.1e" % (
    os.environ.get("W2VS_TN8_DBG", "0"), R, us, fl / us / 1e6, errs, berr), flush=True)
