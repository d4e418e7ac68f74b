#!/usr/bin/env python3
"""Where an attention launch spends its time (tuning build only: make -C wav2vec-s_amd/csrc tuning).
In-kernel s_memrealtime stamps of attn2_fwd / attn2_dq / attn2_dkv (entry / loop entry / loop exit / behind the merge
barrier / end, per wave) at the cfgB shape: when workgroups start, how long a sub-tile takes as a function of what else is
resident, and which workgroup ends the launch.
    python tools/attn_anatomy_probe.py [p_drop]"""
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
lib.w2vs_dbg_attn_stamps.argtypes = [C.c_void_p, C.c_int]
lib.w2vs_dbg_attn_stamps.restype = None
p_drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
B, H, Tp, m, r = 8, 12, 546, 16, 8
N = Tp + (Tp // m) * r
nT_tiles = (N + 31) // 32
forced = int(os.environ.get("W2VS_ATTN_NW", "0"))
NW_Q = forced if forced in (2, 4) else 2          # waves per workgroup of the forward / dQ launches at this shape (longest list 26 <= 32)
NW_K = int(os.environ.get("W2VS_ATTN_DKV_NW", "0")) or 2
stamps = torch.zeros(B * H * nT_tiles * 4 * 8, dtype=torch.int64, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3 * 64 * H, device="cuda", generator=g).to(BF)
dout = torch.randn(B, N, 64 * H, device="cuda", generator=g).to(BF)


def ev_us(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def report(name, s):
    """s: [rows (launch order, longest first), BH, NW, 8] ticks of 10 ns"""
    s = s.astype(np.float64)
    NW = s.shape[2]
    live = s[:, :, 0, 4] > 0
    t0 = s[:, :, :, 0][s[:, :, :, 0] > 0].min()
    u = (s[..., :5] - t0) / 100.0                       # us since the first wave's entry
    nT = s[:, 0, 0, 5].astype(int)
    end = u[:, :, 0, 4]
    print("== %s: launch spans %.1f us (first entry -> last end); %d workgroups" % (name, end[live].max(), live.sum()))
    ent = u[:, :, 0, 0]
    print("   entry times: p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us" % tuple(np.percentile(ent[live], [50, 90, 99, 100])))
    print("   rows (list length: entry p50 | prologue | loop wave0 / wave1 | per sub-tile | barrier wait | merge+store | end p50 / max)")
    for i in list(range(0, len(nT), max(1, len(nT) // 9))) + [len(nT) - 1]:
        e = u[i]
        pro = e[:, 0, 1] - e[:, 0, 0]
        l0, l1 = e[:, 0, 2] - e[:, 0, 1], e[:, 1, 2] - e[:, 1, 1]
        nw_i = max(1, int((s[i, 0, :, 0] > 0).sum()))         # waves of this row that ran
        per = l0 / max(1, (nT[i] + nw_i - 1) // nw_i)
        bar = e[:, 0, 3] - e[:, 0, 2]
        fin = e[:, 0, 4] - e[:, 0, 3]
        print("   nT %2d: entry %5.1f | pro %4.1f | loop %5.1f / %5.1f | %4.2f | bar %4.1f | fin %4.1f | end %5.1f / %5.1f" % (
            nT[i], np.median(e[:, 0, 0]), np.median(pro), np.median(l0), np.median(l1), np.median(per), np.median(bar), np.median(fin),
            np.median(e[:, 0, 4]), e[:, 0, 4].max()))
    # what ends the launch
    idx = np.unravel_index(np.argmax(np.where(live, end, -1)), end.shape)
    print("   last workgroup: row %d (nT %d), entry %.1f, loop entry %.1f, loop exit %.1f, end %.1f" % (
        idx[0], nT[idx[0]], u[idx][0, 0], u[idx][0, 1], u[idx][0, 2], u[idx][0, 4]))
    # residency: waves alive over time
    ts = np.linspace(0, end[live].max(), 13)
    ran = (s[:, :, :, 0] > 0).reshape(-1)
    a0, a1 = u[:, :, :, 0].reshape(-1)[ran], np.maximum(u[:, :, :, 3], u[:, :, :, 4]).reshape(-1)[ran]
    l_in, l_out = u[:, :, :, 1].reshape(-1)[ran], u[:, :, :, 2].reshape(-1)[ran]
    inloop = [int(((l_in <= t) & (l_out > t)).sum()) for t in ts]
    alive = [int(((a0 <= t) & (a1 > t)).sum()) for t in ts]
    print("   t (us):        " + " ".join("%5.1f" % t for t in ts))
    print("   waves alive:   " + " ".join("%5d" % a for a in alive))
    print("   waves in loop: " + " ".join("%5d" % a for a in inloop))
    # placement: workgroups per CU
    hw = s[:, :, 0, 6].astype(np.int64)
    xcc = s[:, :, 0, 7].astype(np.int64) & 15
    cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (xcc << 8)
    ids, cnt = np.unique(cu[live], return_counts=True)
    work = np.zeros(len(ids))
    for j, c in enumerate(ids):
        work[j] = (np.broadcast_to(nT[:, None], cu.shape)[(cu == c) & live]).sum()
    print("   %d distinct (xcc, se, cu) ids; workgroups per id %d-%d; sub-tiles per id min %d p50 %d max %d (mean %.1f)" % (
        len(ids), cnt.min(), cnt.max(), work.min(), np.median(work), work.max(), work.mean()))


for p in (p_drop,):
    o, lse = ops.attn_fwd(qkv, H, Tp, m, r, p_drop=p, seed=5)
    fwd = lambda: ops.attn_fwd(qkv, H, Tp, m, r, p_drop=p, seed=5)        # noqa: E731
    bwd = lambda: ops.attn_bwd(dout, qkv, o, lse, H, Tp, m, r, p_drop=p, seed=5)   # noqa: E731
    lib.w2vs_dbg_attn_stamps(None, 0)
    print("p = %.2f, cfgB shape (N = %d): forward %.1f us, backward (dQ + dK/dV) %.1f us (event-timed, back to back)" % (
        p, N, ev_us(fwd), ev_us(bwd)))
    for pas, name, fn in ((0, "forward", fwd), (1, "dQ pass", bwd), (2, "dK/dV pass", bwd)):
        lib.w2vs_dbg_attn_stamps(stamps.data_ptr(), pas)
        for _ in range(2):
            stamps.zero_()
            torch.cuda.synchronize()
            fn()
            torch.cuda.synchronize()
        nw = NW_K if pas == 2 else NW_Q
        s = stamps.cpu().numpy()[: nT_tiles * B * H * nw * 8].reshape(nT_tiles, B * H, nw, 8)
        report(name, s)
    lib.w2vs_dbg_attn_stamps(None, 0)
