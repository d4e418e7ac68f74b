#!/usr/bin/env python3
"""Where a K = 768 NT GEMM launch spends its time (tuning build only: make -C wav2vec-s_amd/csrc tuning).
In-kernel s_memrealtime stamps of the 8-phase kernel (entry / first K tile landed / end of the K loop / epilogue issued /
stores drained, per wave group and workgroup) and timing-only epilogue ablations (no GELU arithmetic, no gelu' store, no store).
    python tools/nt_anatomy_probe.py [R]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("W2VS_LIB", os.path.join(ROOT, "wav2vec-s_amd", "libw2vs_tuning.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16
lib = _lib.load()
lib.w2vs_dbg_nt_stamps.argtypes = [C.c_void_p, C.c_int]
lib.w2vs_dbg_nt_stamps.restype = None
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6544
g = torch.Generator(device="cuda").manual_seed(0)
stamps = torch.zeros(256 * 2 * 8, dtype=torch.int64, device="cuda")
evict = torch.empty(160 << 20, dtype=torch.float32, device="cuda")      # 640 MB: beyond the 256 MB Infinity Cache


def t_us(fn, iters=20, cold=False):
    for _ in range(3):
        fn()
    tot = 0.0
    for _ in range(iters):
        if cold:
            evict.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3


def anatomy(fn, cold, dbg=0):
    lib.w2vs_dbg_nt_stamps(stamps.data_ptr(), dbg)
    rows = []
    for _ in range(6):
        stamps.zero_()
        if cold:
            evict.fill_(1.0)
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(256, 2, 8).astype(np.float64)
        live = s[:, :, 6] > 0
        if not live.any():
            continue
        t0 = s[:, :, 0][live].min()
        u = (s - t0) / 100.0                      # 100 MHz ticks -> us
        u[~live] = np.nan
        rows.append([np.nanmedian(u[:, :, 0]), np.nanmax(u[:, :, 0]), np.nanmedian(u[:, :, 1] - u[:, :, 0]), np.nanmedian(u[:, :, 2] - u[:, :, 1]),
                     np.nanmedian(u[:, :, 3] - u[:, :, 2]), np.nanmedian(u[:, :, 6] - u[:, :, 3]), np.nanmedian(u[:, :, 6]), np.nanmax(u[:, :, 6]),
                     np.nanmedian(u[:, 0, 2] - u[:, 0, 1]), np.nanmedian(u[:, 1, 2] - u[:, 1, 1]),
                     np.nanmedian(u[:, 0, 3] - u[:, 0, 2]), np.nanmedian(u[:, 1, 3] - u[:, 1, 2])])
    lib.w2vs_dbg_nt_stamps(None, 0)
    r = np.median(np.array(rows[1:]), axis=0)
    return ("start med %.1f max %.1f | first K tile %.1f | K loop %.1f | epilogue issue %.1f | drain %.1f | end med %.1f max %.1f"
            " | per wave group: K loop %.1f / %.1f, epilogue issue %.1f / %.1f" % tuple(r))


for name, N, K in (("fc1 fwd", 3072, 768), ("qkv fwd", 2304, 768), ("fc2 fwd", 768, 3072)):
    x = torch.randn(R, K, device="cuda", generator=g).to(BF)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(BF)
    b = torch.randn(N, device="cuda", generator=g).to(BF)
    aux = torch.randn(R, N, device="cuda", generator=g).to(BF)
    forms = {"bias": lambda: ops.linear_fwd(x, w, b), "gelu+saveg": lambda: ops.linear_fwd(x, w, b, gelu=True, save_pre=True, save_grad=True),
             "mul-aux": lambda: ops.linear_dgrad(x, w, mul_aux=aux)}
    if K != 768:
        ops.gemm_tune(8, 256)                     # force the 8-phase kernel (the stamps live there)
    for fname, fn in forms.items():
        for cold in (False, True):
            print("%-8s %-10s %-4s %6.1f us   %s" % (name, fname, "cold" if cold else "warm", t_us(fn, cold=cold), anatomy(fn, cold)), flush=True)
    if K == 768:
        for dbg, what in ((0, "as shipped"), (1, "no GELU arithmetic"), (2, "no gelu' store"), (3, "neither"), (4, "no store at all"), (5, "no arithmetic, no store")):
            lib.w2vs_dbg_nt_stamps(None, dbg)
            print("%-8s gelu+saveg ablation %d (%s): warm %6.1f us  cold %6.1f us" % (name, dbg, what, t_us(forms["gelu+saveg"]), t_us(forms["gelu+saveg"], cold=True)), flush=True)
        lib.w2vs_dbg_nt_stamps(None, 0)
    ops.gemm_tune(-1, 0)
