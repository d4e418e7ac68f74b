#!/usr/bin/env python3
"""The four weight gradients of an encoder layer as one grouped launch (w2vs_gemm_tn_group) at the cfgB shape.
W2VS_TN8=0|1 selects the 256x128 single-writer kernel or the 8-phase 256x256 split-K kernel (read once per process).
python tools/wgrad_group_probe.py [R]"""
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
dys = [mk(R, 3 * E), mk(R, E), mk(R, F), mk(R, E)]          # d(qkv), d(out_proj out), d(fc1 out), d(fc2 out)
xs = [mk(R, E), mk(R, E), mk(R, E), mk(R, F)]
ws = [torch.zeros(dy.shape[1], x.shape[1], device="cuda") for dy, x in zip(dys, xs)]
bs = [torch.zeros(dy.shape[1], device="cuda") for dy in dys]
prob = [dict(a=dy, b=x, out_f32=w, M=dy.shape[1], N=x.shape[1], K=R, lda=dy.shape[1], ldb=x.shape[1], ldc=x.shape[1],
             alpha=1.0, colsum_out=b) for dy, x, w, b in zip(dys, xs, ws, bs)]
ops.gemm_tn_group(prob)
torch.cuda.synchronize()
errs = [float((w - dy.float().t() @ x.float()).norm() / (dy.float().t() @ x.float()).norm()) for dy, x, w in zip(dys, xs, ws)]
berr = [float((b - dy.float().sum(0)).norm() / dy.float().sum(0).norm()) for dy, b in zip(dys, bs)]
for _ in range(5):
    ops.gemm_tn_group(prob)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.gemm_tn_group(prob)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
fl = sum(2.0 * R * dy.shape[1] * x.shape[1] for dy, x in zip(dys, xs))
print("W2VS_TN8=%s R=%d: %.1f us = %.0f TF/s   rel err dW %s  db %s" % (os.environ.get("W2VS_TN8", "1"), R, us, fl / us / 1e6,
      ["%.1e" % e for e in errs], ["%.1e" % e for e in berr]), flush=True)
