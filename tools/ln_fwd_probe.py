import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wav2vec_s_amd
from wav2vec_s_amd import ops
BF = torch.bfloat16
def t_us(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
R, C = 6544, 768
x = torch.randn(R, C, device="cuda").to(BF); r = torch.randn(R, C, device="cuda").to(BF)
g = torch.ones(C, device="cuda").to(BF); b = torch.zeros(C, device="cuda").to(BF)
y = torch.empty_like(x); s = torch.empty_like(x)
print("torch copy  (10+10 MB)  %.1f us" % t_us(lambda: y.copy_(x)))
print("torch add   (20+10 MB)  %.1f us" % t_us(lambda: torch.add(x, r, out=s)))
print("torch layer_norm        %.1f us" % t_us(lambda: torch.nn.functional.layer_norm(x, (C,), g, b)))
print("w2vs ln_fwd plain (10+10)        %.1f us" % t_us(lambda: ops.ln_fwd(x, g, b)))
print("w2vs ln_fwd res+sum (20+20)      %.1f us" % t_us(lambda: ops.ln_fwd(x, g, b, res=r, want_sum=True)))
print("w2vs ln_fwd res+sum+drop (20+20) %.1f us" % t_us(lambda: ops.ln_fwd(x, g, b, res=r, want_sum=True, p_drop=0.1, seed=3)))
yy, ss, mean, rstd = ops.ln_fwd(x, g, b, res=r, want_sum=True)
dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
dy = torch.randn(R, C, device="cuda").to(BF)
print("w2vs ln_bwd dx (20+10)           %.1f us" % t_us(lambda: ops.ln_bwd(ss, g, b, mean, rstd, dg, db, dy=dy)))
print("w2vs ln_bwd dx+dres dsum (30+20) %.1f us" % t_us(lambda: ops.ln_bwd(ss, g, b, mean, rstd, dg, db, dy=dy, dsum=r, want_dres=True, p_drop=0.1, seed=3)))
