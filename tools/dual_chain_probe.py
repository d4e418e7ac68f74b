#!/usr/bin/env python3
"""Does running the encoder as TWO half-batch chains on two HIP streams (rows [0, R/2) and [R/2, R) of the same tensors, the
weight gradients joint over all R rows on a third stream) beat one full-batch chain?  Encoder layers only, cfgB shape
(B = 8, N = 818, E = 768, F = 3072, 12 layers, post-LN), forward and backward timed separately with events.
python tools/dual_chain_probe.py [B] [Tp] [layers]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wav2vec_s_amd  # noqa: E402,F401
from wav2vec_s_amd import _lib, host_rng, ops  # noqa: E402,F401
from wav2vec_s_amd._lib import LayerDesc  # noqa: E402

BF = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Tp = int(sys.argv[2]) if len(sys.argv) > 2 else 546
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 12
E, F, H, m, r = 768, 3072, 12, 16, 8
N = host_rng.block_layout(Tp, m, r).N
R = B * N
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=0.05: (torch.randn(*s, device=dev, generator=g) * sc).to(BF)      # noqa: E731
P_DROP = 0.1

Ws = []
for j in range(NL):
    w = dict(wqkv=rnd(3 * E, E), bqkv=rnd(3 * E), wo=rnd(E, E), bo=rnd(E), w1=rnd(F, E), b1=rnd(F), w2=rnd(E, F), b2=rnd(E),
             ln1_g=torch.ones(E, device=dev, dtype=BF), ln1_b=torch.zeros(E, device=dev, dtype=BF),
             ln2_g=torch.ones(E, device=dev, dtype=BF), ln2_b=torch.zeros(E, device=dev, dtype=BF))
    w["wqkv_t"], w["wo_t"] = w["wqkv"].t().contiguous(), w["wo"].t().contiguous()
    w["w1_t"], w["w2_t"] = w["w1"].t().contiguous(), w["w2"].t().contiguous()
    w["g"] = {k: torch.zeros(v.shape, device=dev) for k, v in w.items() if not k.endswith("_t")}
    Ws.append(w)
x0 = rnd(R, E, sc=1.0)
per16 = R * (8 * E + 2 * F)
slab16 = torch.empty(NL * per16, device=dev, dtype=BF)
per32 = B * H * N + 4 * R
slab32 = torch.empty(NL * per32, device=dev, dtype=torch.float32)
tmp = torch.empty(2 * R * E, device=dev, dtype=BF)
NSET = 4
sets = [dict(ws_e0=torch.empty(R, E, device=dev, dtype=BF), ws_f=torch.empty(R, F, device=dev, dtype=BF),
             ws_qkv=torch.empty(R, 3 * E, device=dev, dtype=BF), ws_e3=torch.empty(R, E, device=dev, dtype=BF)) for _ in range(NSET)]
ws_e1, ws_e2 = torch.empty(R, E, device=dev, dtype=BF), torch.empty(R, E, device=dev, dtype=BF)
delta = torch.empty(B * H * N, device=dev, dtype=torch.float32)
dbuf = [rnd(R, E, sc=1.0), torch.empty(R, E, device=dev, dtype=BF)]
tn_ws = torch.empty(36 << 18, device=dev, dtype=torch.float32)       # 36 MB


def desc(j, b0, nb, chain):
    """Layer j on the batch rows [b0, b0 + nb): every pointer offset to that row range of the shared tensors."""
    w = Ws[j]
    d = LayerDesc()
    d.B, d.N, d.E, d.F, d.H, d.Tp, d.m, d.r, d.post_ln, d.num_cu = nb, N, E, F, H, Tp, m, r, 1, 256
    d.p_drop, d.p_attn = P_DROP, P_DROP
    d.seed_attn, d.seed_drop1, d.seed_drop2 = 1000 + j + 77 * b0, 2000 + j + 77 * b0, 3000 + j + 77 * b0
    for k in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b", "wqkv_t", "wo_t", "w1_t", "w2_t"):
        setattr(d, k, w[k].data_ptr())
    r0 = b0 * N
    base16 = slab16.data_ptr() + 2 * j * per16
    o = 0
    for f_, cols in (("qkv", 3 * E), ("ctx", E), ("s1", E), ("x1", E), ("hpre", F), ("h", F), ("s2", E), ("x_out", E)):
        setattr(d, f_, base16 + 2 * (o + r0 * cols))
        o += R * cols
    base32 = slab32.data_ptr() + 4 * j * per32
    d.lse = base32 + 4 * (b0 * H * N)
    d.mean1, d.rstd1 = base32 + 4 * (B * H * N + r0), base32 + 4 * (B * H * N + R + r0)
    d.mean2, d.rstd2 = base32 + 4 * (B * H * N + 2 * R + r0), base32 + 4 * (B * H * N + 3 * R + r0)
    if j == 0:
        d.x_in = x0.data_ptr() + 2 * r0 * E
    else:
        d.x_in = slab16.data_ptr() + 2 * ((j - 1) * per16 + R * (7 * E + 2 * F) + r0 * E)
    d.tmp = tmp.data_ptr() + 2 * (chain * R * E + 0)
    # backward
    jj = NL - 1 - j
    d.d_out = dbuf[jj & 1].data_ptr() + 2 * r0 * E
    d.d_in = dbuf[(jj + 1) & 1].data_ptr() + 2 * r0 * E
    s = sets[jj % NSET]
    d.ws_e0, d.ws_f = s["ws_e0"].data_ptr() + 2 * r0 * E, s["ws_f"].data_ptr() + 2 * r0 * F
    d.ws_qkv, d.ws_e3 = s["ws_qkv"].data_ptr() + 2 * r0 * 3 * E, s["ws_e3"].data_ptr() + 2 * r0 * E
    d.ws_e1, d.ws_e2 = ws_e1.data_ptr() + 2 * r0 * E, ws_e2.data_ptr() + 2 * r0 * E
    d.delta = delta.data_ptr() + 4 * b0 * H * N
    half = tn_ws.numel() * 4 // 2
    d.tn_ws, d.tn_ws_bytes = (tn_ws.data_ptr() + chain * half, half) if nb < B else (tn_ws.data_ptr(), tn_ws.numel() * 4)
    for k in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b"):
        setattr(d, "g_" + k, w["g"][k].data_ptr())
    d.defer_wgrads = 1
    return d


full = [desc(j, 0, B, 0) for j in range(NL)]
hb = B // 2
halves = [[desc(j, 0, hb, 0) for j in range(NL)], [desc(j, hb, B - hb, 1) for j in range(NL)]]
cur = torch.cuda.current_stream()
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
sp = lambda s: C.c_void_p(s.cuda_stream)      # noqa: E731


def fwd_single():
    for j in range(NL):
        _lib.call("w2vs_layer_fwd", C.byref(full[j]), sp(cur))


def fwd_dual():
    ev = torch.cuda.Event(); ev.record(cur)
    s0.wait_event(ev); s1.wait_event(ev)
    for j in range(NL):
        _lib.call("w2vs_layer_fwd", C.byref(halves[0][j]), sp(s0))
        _lib.call("w2vs_layer_fwd", C.byref(halves[1][j]), sp(s1))
    for s in (s0, s1):
        e = torch.cuda.Event(); e.record(s); cur.wait_event(e)


def wg(pair, stream):
    arr = (LayerDesc * len(pair))(*pair)
    _lib.call("w2vs_layer_wgrads", arr, len(pair), sp(stream))


def bwd_single():
    pend = None
    for j in reversed(range(NL)):
        _lib.call("w2vs_layer_bwd", C.byref(full[j]), sp(cur))
        if pend is None:
            pend = full[j]
        else:
            wg([pend, full[j]], cur)
            pend = None
    if pend is not None:
        wg([pend], cur)


def bwd_dual(wg_stream_mode):
    """wg_stream_mode: 'side' = the joint weight gradients on a third stream, 'inline' = on chain 0's stream after a join"""
    ev = torch.cuda.Event(); ev.record(cur)
    s0.wait_event(ev); s1.wait_event(ev); s2.wait_event(ev)
    pend, wg_done = None, {}
    for j in reversed(range(NL)):
        jj = NL - 1 - j
        if jj - NSET in wg_done:                      # the operand set is about to be overwritten: its readers must be done
            s0.wait_event(wg_done[jj - NSET]); s1.wait_event(wg_done[jj - NSET])
        _lib.call("w2vs_layer_bwd", C.byref(halves[0][j]), sp(s0))
        _lib.call("w2vs_layer_bwd", C.byref(halves[1][j]), sp(s1))
        if pend is None:
            pend = (j, jj)
        else:
            ws = s2 if wg_stream_mode == "side" else s0
            e0, e1 = torch.cuda.Event(), torch.cuda.Event()
            e0.record(s0); e1.record(s1)
            ws.wait_event(e1)
            if ws is not s0:
                ws.wait_event(e0)
            wg([full[pend[0]], full[j]], ws)
            e = torch.cuda.Event(); e.record(ws)
            wg_done[pend[1]] = e; wg_done[jj] = e
            pend = None
    if pend is not None:
        ws = s2 if wg_stream_mode == "side" else s0
        e0, e1 = torch.cuda.Event(), torch.cuda.Event()
        e0.record(s0); e1.record(s1)
        ws.wait_event(e1)
        if ws is not s0:
            ws.wait_event(e0)
        wg([full[pend[0]]], ws)
    for s in (s0, s1, s2):
        e = torch.cuda.Event(); e.record(s); cur.wait_event(e)


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(cur)
        fn()
        b.record(cur)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


print("B %d N %d (Tp %d) R %d, %d layers, dropouts %.1f" % (B, N, Tp, R, NL, P_DROP), flush=True)
fwd_single(); torch.cuda.synchronize()
ref_out = slab16[(NL - 1) * per16 + R * (7 * E + 2 * F):][:R * E].clone()
fwd_dual(); torch.cuda.synchronize()
dual_out = slab16[(NL - 1) * per16 + R * (7 * E + 2 * F):][:R * E]
# the second half's dropout seeds differ from the full launch's, so only the first half's rows must match bit for bit
print("first-half rows identical:", bool(torch.equal(ref_out[:hb * N * E], dual_out[:hb * N * E])), flush=True)
for name, fn in (("fwd single", fwd_single), ("fwd dual", fwd_dual), ("fwd single", fwd_single), ("fwd dual", fwd_dual)):
    med, best = timeit(fn)
    print("%-22s median %.3f ms  best %.3f ms" % (name, med, best), flush=True)
fwd_single(); torch.cuda.synchronize()
for name, fn in (("bwd single", bwd_single), ("bwd dual wg-side", lambda: bwd_dual("side")), ("bwd dual wg-inline", lambda: bwd_dual("inline")),
                 ("bwd single", bwd_single), ("bwd dual wg-side", lambda: bwd_dual("side")), ("bwd dual wg-inline", lambda: bwd_dual("inline"))):
    med, best = timeit(fn)
    print("%-22s median %.3f ms  best %.3f ms" % (name, med, best), flush=True)
