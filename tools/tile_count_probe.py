#!/usr/bin/env python3
"""How does the duration of ONE grouped weight-gradient launch (gemm_tn8_group_kernel, 256 x 256 tiles, full-length K loops, no
split) depend on its tile count?  n problems of 36 tiles each (dW[3072, 768] = dY[R, 3072]^T X[R, 768], R = 6544) for
n = 2 .. 7 (72 .. 252 tiles), operands rotated over `sets` copies so that they come from HBM / the Infinity Cache as in the step.
python tools/tile_count_probe.py [R] [sets]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import _lib, ops  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6544
SETS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
E, F = 768, 3072
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).to(torch.bfloat16)      # noqa: E731
pool = [[(mk(R, F), mk(R, E), torch.zeros(F, E, device=dev)) for _ in range(7)] for _ in range(SETS)]
_lib.call("w2vs_gemm_tn8_max_split", 1)
for n in (2, 3, 4, 5, 6, 7):
    def run(k):
        ops.gemm_tn_group([dict(a=dy, b=x, out_f32=dw, M=F, N=E, K=R, lda=F, ldb=E, ldc=E, overwrite=1) for dy, x, dw in pool[k % SETS][:n]])
    for k in range(3):
        run(k)
    torch.cuda.synchronize()
    form = _lib.load().w2vs_gemm_last_group_form()
    ts = []
    for k in range(24):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run(k)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    med = ts[len(ts) // 2]
    fl = 2.0 * R * E * F * n
    print("%d problems = %3d tiles (form %d): median %.1f us, best %.1f us  = %.0f TF/s  (%.2f us per tile-slot of 256 CUs: %.1f)" %
          (n, 36 * n, form, med, ts[0], fl / med / 1e6, med / (36 * n), med), flush=True)
