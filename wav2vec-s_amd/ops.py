"""Tensor-level wrappers of the libw2vs kernels (one Python function per C-ABI entry).

torch is used here only for device memory and the current HIP stream; every computation is a
hand-written HIP kernel.  Nothing falls back to torch ops or to the CPU: a missing library or a
rejected call raises ``W2vsError``.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (EPI_ADD, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_SAVE, EPI_BIAS_GELU_SAVEG, EPI_DGELU, EPI_F32, EPI_MUL, EPI_NONE, AttnDesc,
                   EncPrologueDesc, GemmDesc, InfonceLossDesc, LnBwdDesc, LnFwdDesc, NceDesc, QuantDesc, W2vsError)

BF16 = torch.bfloat16


class _StepArena:
    """Bump allocator for the activations / scratch of one training step.

    Sampled block contexts change the token count N every step; through torch's caching allocator
    that means fresh multi-GB hipMalloc/hipFree pairs (device-synchronising) per step.  When active,
    every buffer the kernels write comes from ONE preallocated slab that is rewound at step start -
    288 GB of HBM3E makes reserving the worst case trivial.  Inactive (default, and always outside
    ``trainer.TrainStep.__call__``): plain torch.empty."""

    def __init__(self):
        self.buf, self.off, self.cap, self.active = None, 0, 0, False

    def activate(self, nbytes, device):
        """Reserve the slab (kept across steps) and start handing out from it."""
        if self.buf is None or self.cap < nbytes or self.buf.device != torch.device(device):
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self.cap = nbytes
        self.off = 0
        self.active = True

    def suspend(self):
        """Keep the slab, stop handing out from it: outside a training step (validation forwards, the streaming twin,
        a second model in the process) buffers come from torch's allocator and cannot be overwritten by the next step."""
        self.active = False

    def deactivate(self):
        self.buf, self.off, self.cap, self.active = None, 0, 0, False

    def reset(self):
        self.off = 0

    def empty(self, shape, dtype, device):
        if isinstance(shape, int):
            shape = (shape,)
        if not self.active or self.buf is None:
            return torch.empty(shape, device=device, dtype=dtype)
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * _ITEMSIZE[dtype]
        start = (self.off + 255) & ~255
        if start + nbytes > self.cap:
            return torch.empty(shape, device=device, dtype=dtype)   # overflow: fall back to the allocator
        self.off = start + nbytes
        return self.buf[start:start + nbytes].view(dtype).view(shape)


_ITEMSIZE = {torch.bfloat16: 2, torch.float32: 4, torch.int32: 4, torch.int64: 8, torch.uint8: 1, torch.float16: 2}
ARENA = _StepArena()


def empty(shape, dtype, device):
    return ARENA.empty(shape, dtype, device)


def zeros(shape, dtype, device):
    t = ARENA.empty(shape, dtype, device)
    t.zero_()
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream of the current device as a raw handle.  torch.cuda.current_stream() builds a Stream object
    (~10 us of host time per call: 1 ms of a launch-bound step with 500 launches); the raw getter is a plain C call."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _chk(t, dtype=None, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        raise W2vsError("%s must live on the GPU (no CPU path exists)" % name)
    if dtype is not None and t.dtype != dtype:
        raise W2vsError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise W2vsError("%s must be contiguous" % name)


# ------------------------------------------------------------------------------------------ GEMM
class _GemmTimer:
    """bench.py roofline: libw2vs brackets every stride-th GEMM launch (Python- or C++-issued alike) with
    HIP events on the launching stream; totals are read back after the timed region.  The stride is a prime
    that shares no factor with the 4 / 8 GEMMs of a layer's forward / backward, so successive samples walk
    through every launch position instead of re-timing the same two kernels each step; an event pair costs
    ~12 us of stream time, so 1 in 29 keeps the probe below 0.5 % of the step."""

    def __init__(self, stride=29):
        self.stride = stride

    def enable(self):
        _lib.call("w2vs_prof_enable", self.stride)

    def disable(self):
        pass  # samples are kept until the next enable()

    def begin(self):
        return None

    def end(self, *a):
        return None

    def report(self, peak_tflops):
        torch.cuda.synchronize()
        # one id per kernel symbol family: NT = 16 * form + epilogue, the weight-gradient (TN) kernels 10..12
        epis = ["none", "bias", "bias_gelu", "bias_gelu_save", "dgelu", "f32", "add", "bias_gelu_savegrad", "mul"]
        names = {16 * f + e: "%s<%s>" % (fn, en) for f, fn in enumerate(("gemm_nt_kernel", "gemm_nt_lc_kernel", "gemm_nt_p_kernel", "gemm_nt8_kernel"))
                 for e, en in enumerate(epis)}
        names.update({10: "gemm_tn_kernel", 11: "gemm_tn_lc_kernel (+ tn_slab_reduce_kernel)", 12: "gemm_tn_group_kernel",
                      13: "gemm_tn8_group_kernel"})
        groups, raw = {}, {}
        for k in names:
            ms, fl, n = C.c_double(), C.c_double(), C.c_int32()
            _lib.call("w2vs_prof_read", k, C.byref(ms), C.byref(fl), C.byref(n))
            if n.value:
                groups[k] = (ms.value * 1e-3, fl.value, n.value, int(_lib.load().w2vs_prof_launches(k)),
                             float(_lib.load().w2vs_prof_flops(k)))
                _lib.call("w2vs_prof_read_raw", k, C.byref(ms), C.byref(fl), C.byref(n))
                raw[k] = (ms.value * 1e-3, fl.value, n.value)
        _lib.call("w2vs_prof_enable", 0)
        if not groups:
            return None
        # dominant = largest ESTIMATED total, not largest sampled total (a 1-in-29 sample of a kernel launched 10 times a step is
        # too thin to rank by): the sample's time scaled by (FLOPs of ALL launches of the family / FLOPs of the timed ones) - a
        # family mixes shapes (the fc2 dgrad and the conv dgrads share a symbol), so scaling by launch COUNT mis-ranked it
        # whenever the sample happened to hold the big ones
        k, (t, fl, n, _, _) = max(groups.items(), key=lambda kv: kv[1][0] * max(kv[1][4], kv[1][1]) / max(kv[1][1], 1.0))
        ach = fl / t / 1e12
        out = {"bound": "mfma", "kernel": names[k], "achieved": round(ach, 1), "peak": peak_tflops, "unit": "TFLOP/s",
               "frac": round(ach / peak_tflops, 4), "traffic": None, "launches_timed": n,
               "avg_launch_us": round(t / n * 1e6, 2), "flops_per_launch_avg": round(fl / n / 1e9, 3),
               "launches_in_timed_region": groups[k][3]}
        # the samples the host-gap filter left out (time per FLOP > 3 x the family median), and what the totals are with them
        out["samples_dropped"] = raw[k][2] - n
        out["achieved_unfiltered"] = round(raw[k][1] / raw[k][0] / 1e12, 1)
        for tag, sel in (("nt", lambda kk: kk not in (10, 11, 12, 13)), ("tn", lambda kk: kk in (10, 11, 12, 13))):
            gs = [g for kk, g in groups.items() if sel(kk)]
            rs = [g for kk, g in raw.items() if sel(kk)]
            if gs:
                out["all_gemm_%s_tflops" % tag] = round(sum(g[1] for g in gs) / sum(g[0] for g in gs) / 1e12, 1)
                out["all_gemm_%s_tflops_unfiltered" % tag] = round(sum(g[1] for g in rs) / sum(g[0] for g in rs) / 1e12, 1)
                out["all_gemm_%s_samples_dropped" % tag] = sum(g[2] for g in rs) - sum(g[2] for g in gs)
        return out


GEMM_TIMER = _GemmTimer()


def gemm_nt(a, b, *, M, N, K, lda, ldb, ldc, out=None, out2=None, out_f32=None, bias=None, aux=None,
            epi=EPI_NONE, a_off=0, batch=1, sA=0, sB=0, sC=0, a_bytes=0, b_bytes=0, c_elems=0, alpha=1.0, zk_col=0, zk_k=0):
    d = GemmDesc()
    d.A, d.B, d.C, d.C2, d.Cf, d.bias, d.aux = _p(a), _p(b), _p(out), _p(out2), _p(out_f32), _p(bias), _p(aux)
    d.M, d.N, d.K, d.batch = M, N, K, batch
    d.lda, d.ldb, d.ldc, d.a_off = lda, ldb, ldc, a_off
    d.sA, d.sB, d.sC = sA, sB, sC
    d.a_bytes, d.b_bytes, d.c_elems = a_bytes, b_bytes, c_elems
    d.epi, d.alpha = epi, alpha
    d.zk_col, d.zk_k = zk_col, zk_k
    ev = GEMM_TIMER.begin()
    _lib.call("w2vs_gemm_nt", C.byref(d), _stream())
    GEMM_TIMER.end(ev, epi, 2.0 * M * N * K * batch)


def gemm_tune(nt_mode=-1, lc_height=0, tn_lc=-1):
    """Force a GEMM kernel variant (tests / tuning); defaults restore the automatic choice."""
    _lib.call("w2vs_gemm_tune", nt_mode, lc_height, tn_lc)


_TN_WS = {}


def tn_workspace(device):
    """One persistent 36 MB scratch per device for the split-K partial tiles of the weight-gradient GEMMs (launches
    on a stream are ordered, so every wgrad of a step can share it)."""
    key = str(device)
    if key not in _TN_WS:
        _TN_WS[key] = torch.empty(9 * 1024 * 1024, dtype=torch.float32, device=device)
    return _TN_WS[key]


def gemm_tn(a, b, out_f32, *, M, N, K, lda, ldb, ldc, a_off=0, batch=1, sA=0, sB=0, a_bytes=0, b_bytes=0,
            alpha=1.0, num_cu=256, colsum_out=None):
    d = GemmDesc()
    d.A, d.B, d.Cf = _p(a), _p(b), _p(out_f32)
    d.M, d.N, d.K, d.batch = M, N, K, batch
    d.lda, d.ldb, d.ldc, d.a_off = lda, ldb, ldc, a_off
    d.sA, d.sB = sA, sB
    d.a_bytes, d.b_bytes = a_bytes, b_bytes
    d.alpha = alpha
    d.colsum = _p(colsum_out)
    ws = tn_workspace(a.device)
    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
    ev = GEMM_TIMER.begin()
    _lib.call("w2vs_gemm_tn", C.byref(d), num_cu, _stream())
    GEMM_TIMER.end(ev, "tn", 2.0 * M * N * K * batch)


def gemm_tn_group(problems, num_cu=256):
    """problems: up to 12 dicts with the keyword arguments of ``gemm_tn`` (a, b, out_f32, M, N, K, lda, ldb, ldc[, alpha,
    colsum_out]): weight-gradient GEMMs sharing the reduction dimension, enqueued as ONE launch (w2vs_gemm_tn_group)."""
    n = len(problems)
    arr = (GemmDesc * n)()
    flops = 0.0
    ws = tn_workspace(problems[0]["a"].device)
    for d, pr in zip(arr, problems):
        d.A, d.B, d.Cf = _p(pr["a"]), _p(pr["b"]), _p(pr["out_f32"])
        d.M, d.N, d.K, d.batch = pr["M"], pr["N"], pr["K"], 1
        d.lda, d.ldb, d.ldc = pr["lda"], pr["ldb"], pr["ldc"]
        d.alpha = pr.get("alpha", 1.0)
        d.colsum = _p(pr.get("colsum_out"))
        d.overwrite = int(pr.get("overwrite", 0))
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4       # only used if the group falls back to single launches
        flops += 2.0 * d.M * d.N * d.K
    ev = GEMM_TIMER.begin()
    _lib.call("w2vs_gemm_tn_group", arr, n, num_cu, _stream())
    GEMM_TIMER.end(ev, "tn", flops)


def linear_fwd(x, w, bias=None, *, gelu=False, save_pre=False, save_grad=False):
    """y = x @ w.T (+ bias) [gelu].  x [R, K] bf16, w [N, K] bf16.  Returns y (and pre if save_pre; with save_grad the
    second tensor is gelu'(pre) instead, what a backward pass multiplies with)."""
    _chk(x, BF16, "x"); _chk(w, BF16, "w"); _chk(bias, BF16, "bias")
    R, K = x.shape
    N = w.shape[0]
    y = empty((R, N), BF16, x.device)
    pre = empty(y.shape, y.dtype, y.device) if save_pre else None
    epi = EPI_BIAS if not gelu else (EPI_BIAS_GELU_SAVE if save_pre else EPI_BIAS_GELU)
    if gelu and save_pre and save_grad:
        epi = EPI_BIAS_GELU_SAVEG
    gemm_nt(x, w, M=R, N=N, K=K, lda=K, ldb=K, ldc=N, out=y, out2=pre, bias=bias, epi=epi)
    return (y, pre) if save_pre else y


def linear_dgrad(dy, w_t, *, dgelu_aux=None, add_aux=None, mul_aux=None):
    """dx = dy @ w  given w_t = w.T contiguous ([K, N]).  Optionally dx *= gelu'(aux), dx *= mul_aux or dx += add_aux."""
    _chk(dy, BF16, "dy"); _chk(w_t, BF16, "w_t"); _chk(dgelu_aux, BF16, "aux"); _chk(add_aux, BF16, "aux"); _chk(mul_aux, BF16, "aux")
    R, N = dy.shape
    K = w_t.shape[0]
    dx = empty((R, K), BF16, dy.device)
    epi, aux = EPI_NONE, None
    if dgelu_aux is not None:
        epi, aux = EPI_DGELU, dgelu_aux
    elif add_aux is not None:
        epi, aux = EPI_ADD, add_aux
    elif mul_aux is not None:
        epi, aux = EPI_MUL, mul_aux
    gemm_nt(dy, w_t, M=R, N=K, K=N, lda=N, ldb=N, ldc=K, out=dx, aux=aux, epi=epi)
    return dx


def linear_wgrad(dy, x, dw_f32, alpha=1.0, db_f32=None):
    """dw[N, K] += alpha * dy[R, N]^T @ x[R, K]; optionally db[N] += alpha * colsum(dy)  (fp32 atomics)."""
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(dw_f32, torch.float32, "dw"); _chk(db_f32, torch.float32, "db")
    R, N = dy.shape
    K = x.shape[1]
    gemm_tn(dy, x, dw_f32, M=N, N=K, K=R, lda=N, ldb=K, ldc=K, alpha=alpha, colsum_out=db_f32)


def colsum(x, out_f32):
    _chk(x, BF16, "x"); _chk(out_f32, torch.float32, "out")
    M, N = x.shape
    _lib.call("w2vs_colsum", _p(x), _p(out_f32), M, N, N, _stream())



def transpose_multi(items):
    """items: (in_ptr, out_ptr, R, C[, ld_in, ld_out]) tuples (device addresses of bf16 matrices, row strides in
    elements, 0 = dense); out[C, R] = in[R, C]^T for all of them in one launch per 64 items."""
    from ._lib import TransposeItem
    for i in range(0, len(items), 64):
        chunk = items[i:i + 64]
        arr = (TransposeItem * len(chunk))()
        for a, it in zip(arr, chunk):
            a.inp, a.out, a.R, a.C = it[0], it[1], it[2], it[3]
            a.ld_in, a.ld_out = (it[4], it[5]) if len(it) > 4 else (0, 0)
        _lib.call("w2vs_transpose_multi", arr, len(chunk), _stream())


def transpose2d(x, batch=1):
    """[batch, R, C] -> [batch, C, R] (bf16)."""
    _chk(x, BF16, "x")
    R, Cc = x.shape[-2], x.shape[-1]
    out = empty((*x.shape[:-2], Cc, R), BF16, x.device)
    _lib.call("w2vs_transpose2d", _p(x), _p(out), R, Cc, batch, _stream())
    return out


def f32_to_bf16(x, scale=1.0, out=None):
    _chk(x, torch.float32, "x"); _chk(out, BF16, "out")
    if out is None:
        out = empty((x.shape), BF16, x.device)
    _lib.call("w2vs_f32_to_bf16", _p(x), _p(out), x.numel(), scale, _stream())
    return out


def bf16_to_f32(x, out):
    """out[i] = float(x[i]): the unpack half of a bf16-compressed gradient exchange."""
    _chk(x, BF16, "x"); _chk(out, torch.float32, "out")
    _lib.call("w2vs_bf16_to_f32", _p(x), _p(out), x.numel(), _stream())
    return out


# -------------------------------------------------------------------------------- conv as GEMM
def conv_pack_weight(w):
    """[Cout, Cin, k] -> [Cout, k*Cin] (tap-major) so a channel-last window is one GEMM row."""
    _chk(w, BF16, "w")
    Cout, Cin, k = w.shape
    return transpose2d(w.view(Cout, Cin, k), batch=Cout).view(Cout, k * Cin)


def conv_cl_fwd(x, w2, k, s, bias=None, *, gelu=True, save_pre=True, save_grad=False):
    """Channel-last Conv1d (no padding): x [B, Lin, Cin], w2 [Cout, k*Cin] -> [B, Lout, Cout].  save_grad: the second
    output is gelu'(pre) instead of pre."""
    _chk(x, BF16, "x"); _chk(w2, BF16, "w2")
    B, Lin, Cin = x.shape
    Cout = w2.shape[0]
    Lout = (Lin - k) // s + 1
    y = empty((B, Lout, Cout), BF16, x.device)
    pre = empty(y.shape, y.dtype, y.device) if save_pre else None
    if gelu:
        epi = (EPI_BIAS_GELU_SAVEG if save_grad else EPI_BIAS_GELU_SAVE) if save_pre else EPI_BIAS_GELU
    else:
        epi = EPI_BIAS
    gemm_nt(x, w2, M=Lout, N=Cout, K=k * Cin, lda=s * Cin, ldb=k * Cin, ldc=Cout, out=y, out2=pre, bias=bias,
            epi=epi, batch=B, sA=Lin * Cin, sC=Lout * Cout, a_bytes=Lin * Cin * 2)
    return (y, pre) if save_pre else y


_DGRAD_W = {}
CONV_DGRAD_SKIP = os.environ.get("W2VS_CONV_DGRAD_SKIP", "1") != "0"    # A/B: 0 = multiply with the structural zero block too


def conv_dgrad_weight_items(key, w2, k, s):
    """Transpose items that build the B operand of conv_cl_dgrad for one conv layer into a persistent buffer (its
    structural zeros are written once): returns (buffer, items).  All layers' items go into ONE transpose_multi."""
    Cout, kc = w2.shape
    Cin = kc // k
    if (k, s) == (2, 2):
        buf = _DGRAD_W.get(key)
        if buf is None or buf.shape != (2 * Cin, Cout):
            buf = _DGRAD_W[key] = torch.empty((2 * Cin, Cout), dtype=BF16, device=w2.device)
        return buf, [(w2.data_ptr(), buf.data_ptr(), Cout, 2 * Cin)]
    if (k, s) == (3, 2):
        buf = _DGRAD_W.get(key)
        if buf is None or buf.shape != (2, Cin, 2, Cout):
            buf = _DGRAD_W[key] = torch.zeros((2, Cin, 2, Cout), dtype=BF16, device=w2.device)
        e = 2  # bytes
        base, src = buf.data_ptr(), w2.data_ptr()
        # tap j of the packed weight is the [Cout, Cin] block at columns j*Cin (row stride 3*Cin); its transpose lands
        # at rows (h, ci), column block g of bt[h][ci][g][co] (row stride 2*Cout): (h,g) = (0,0)<-W2, (0,1)<-W0, (1,1)<-W1
        items = []
        for j, (h, g) in ((2, (0, 0)), (0, (0, 1)), (1, (1, 1))):
            items.append((src + e * j * Cin, base + e * ((h * Cin) * 2 * Cout + g * Cout), Cout, Cin, 3 * Cin, 2 * Cout))
        return buf, items
    raise W2vsError("conv dgrad is built for (k,s) in {(2,2),(3,2)}; got (%d,%d)" % (k, s))


def conv_cl_dgrad(dy, w2, k, s, Lin, *, dgelu_aux=None, mul_aux=None, wprep=None):
    """Gradient wrt the channel-last input of conv_cl_fwd.  dy [B, Lout, Cout]; w2 [Cout, k*Cin] packed.
    (k, s) = (2, 2): non-overlapping windows -> plain GEMM into [B, Lout, 2*Cin].
    (k, s) = (3, 2): input rows pair up, pair p = [dy[p-1] | dy[p]] @ [[W2, 0], [W0, W1]]."""
    _chk(dy, BF16, "dy"); _chk(w2, BF16, "w2")
    B, Lout, Cout = dy.shape
    Cin = w2.shape[1] // k
    dx = empty((B, Lin, Cin), BF16, dy.device)
    epi = EPI_DGELU if dgelu_aux is not None else EPI_NONE
    if mul_aux is not None:          # mul_aux = gelu'(pre) saved by the producing layer's forward
        epi, dgelu_aux = EPI_MUL, mul_aux
    if wprep is not None:
        wt = bt = wprep
    else:
        wt = transpose2d(w2)  # [k*Cin, Cout]: row (j*Cin + ci) = w[:, ci, j]
    if (k, s) == (2, 2):
        if Lin > 2 * Lout:
            dx[:, 2 * Lout:].zero_()
        gemm_nt(dy, wt, M=Lout, N=2 * Cin, K=Cout, lda=Cout, ldb=Cout, ldc=2 * Cin, out=dx, aux=dgelu_aux, epi=epi,
                batch=B, sA=Lout * Cout, sC=Lin * Cin, a_bytes=Lout * Cout * 2, c_elems=Lin * Cin)
    elif (k, s) == (3, 2):
        # B operand [N = 2*Cin, K = 2*Cout]: row-half 0 = [W2^T | W0^T], row-half 1 = [0 | W1^T]
        # (K index 0..Cout-1 multiplies dy[p-1], Cout..2Cout-1 multiplies dy[p])
        if wprep is None:
            wt3 = wt.view(3, Cin, Cout)
            bt = zeros((2, Cin, 2, Cout), BF16, dy.device)
            bt[0, :, 0] = wt3[2]
            bt[0, :, 1] = wt3[0]
            bt[1, :, 1] = wt3[1]
        P = (Lin + 1) // 2
        # the [0 | W1^T] row half: output columns >= Cin never see K indices < Cout (zk_*: those tiles skip half their K loop)
        zk = dict(zk_col=Cin, zk_k=Cout) if (CONV_DGRAD_SKIP and Cin % 128 == 0 and Cout % 128 == 0) else {}
        gemm_nt(dy, bt, M=P, N=2 * Cin, K=2 * Cout, lda=Cout, ldb=2 * Cout, ldc=2 * Cin, out=dx, aux=dgelu_aux,
                epi=epi, a_off=-Cout, batch=B, sA=Lout * Cout, sC=Lin * Cin, a_bytes=Lout * Cout * 2,
                c_elems=Lin * Cin, **zk)
    else:
        raise W2vsError("conv dgrad is built for (k,s) in {(2,2),(3,2)}; got (%d,%d)" % (k, s))
    return dx


def conv_cl_wgrad(dy, x, k, s, dw2_f32, alpha=1.0, db_f32=None):
    """dw2[Cout, k*Cin] += sum_{b,t} dy[b,t,:]^T x_window[b,t,:]  (fp32 accumulate)."""
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(dw2_f32, torch.float32, "dw2")
    B, Lout, Cout = dy.shape
    _, Lin, Cin = x.shape
    gemm_tn(dy, x, dw2_f32, M=Cout, N=k * Cin, K=Lout, lda=Cout, ldb=s * Cin, ldc=k * Cin, batch=B,
            sA=Lout * Cout, sB=Lin * Cin, a_bytes=Lout * Cout * 2, b_bytes=Lin * Cin * 2, alpha=alpha,
            colsum_out=db_f32)


# ---------------------------------------------------------------------------------- conv0
def conv0_fwd(wave, w, ln_w, ln_b, k, s, conv_bias=None):
    _chk(wave, BF16, "wave"); _chk(w, BF16, "w"); _chk(ln_w, BF16, "ln_w"); _chk(ln_b, BF16, "ln_b")
    B, L = wave.shape
    Cc = w.shape[0]
    L0 = (L - k) // s + 1
    y = empty((B, L0, Cc), BF16, wave.device)
    mean = empty((B * L0), torch.float32, wave.device)
    rstd = empty(mean.shape, mean.dtype, mean.device)
    _lib.call("w2vs_conv0_fwd", _p(wave), _p(w), _p(conv_bias), _p(ln_w), _p(ln_b), _p(y), _p(mean), _p(rstd),
              B, L, Cc, k, s, _stream())
    return y, mean, rstd


def conv0_bwd(wave, w, ln_w, ln_b, mean, rstd, dy, k, s, dw, dln_w, dln_b, conv_bias=None, dconv_bias=None):
    _chk(dy, BF16, "dy")
    B, L = wave.shape
    Cc = w.shape[0]
    _lib.call("w2vs_conv0_bwd", _p(wave), _p(w), _p(conv_bias), _p(ln_w), _p(ln_b), _p(mean), _p(rstd), _p(dy),
              _p(dw), _p(dconv_bias), _p(dln_w), _p(dln_b), B, L, Cc, k, s, _stream())


def conv0_gn_fwd(wave, w, gn_w, gn_b, k, s, conv_bias=None):
    _chk(wave, BF16, "wave"); _chk(w, BF16, "w"); _chk(gn_w, BF16, "gn_w"); _chk(gn_b, BF16, "gn_b")
    B, L = wave.shape
    Cc = w.shape[0]
    L0 = (L - k) // s + 1
    y = empty((B, L0, Cc), BF16, wave.device)
    stat = empty((B, Cc, 2), torch.float32, wave.device)
    _lib.call("w2vs_conv0_gn_fwd", _p(wave), _p(w), _p(conv_bias), _p(gn_w), _p(gn_b), _p(y), _p(stat), B, L, Cc, k, s,
              _stream())
    return y, stat


def conv0_gn_bwd(wave, w, gn_w, gn_b, stat, dy, k, s, dw, dgn_w, dgn_b, conv_bias=None, dconv_bias=None):
    _chk(dy, BF16, "dy")
    B, L = wave.shape
    Cc = w.shape[0]
    bstat = empty((B, Cc, 2), torch.float32, wave.device)
    _lib.call("w2vs_conv0_gn_bwd", _p(wave), _p(w), _p(conv_bias), _p(gn_w), _p(gn_b), _p(stat), _p(dy), _p(bstat),
              _p(dw), _p(dconv_bias), _p(dgn_w), _p(dgn_b), B, L, Cc, k, s, _stream())


# ------------------------------------------------------------------------------ LayerNorm rows
def ln_fwd(x, gamma, beta, *, res=None, want_y=True, want_sum=False, sumsq=None, gelu=False, p_drop=0.0, seed=0):
    _chk(x, BF16, "x"); _chk(res, BF16, "res"); _chk(gamma, BF16, "gamma"); _chk(beta, BF16, "beta")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = empty(x.shape, x.dtype, x.device) if want_y else None
    s_out = empty(x.shape, x.dtype, x.device) if want_sum else None
    mean = empty((rows), torch.float32, x.device) if want_y else None
    rstd = empty(mean.shape, mean.dtype, mean.device) if want_y else None
    d = LnFwdDesc()
    d.x, d.res, d.gamma, d.beta, d.y, d.sum_out = _p(x), _p(res), _p(gamma), _p(beta), _p(y), _p(s_out)
    d.mean, d.rstd, d.sumsq = _p(mean), _p(rstd), _p(sumsq)
    d.rows, d.C, d.gelu, d.p_drop, d.seed = rows, Cc, int(gelu), p_drop, seed
    _lib.call("w2vs_ln_fwd", C.byref(d), _stream())
    return y, s_out, mean, rstd


def ln_bwd(x, gamma, beta, mean, rstd, dgamma, dbeta, *, dy=None, dsum=None, aux=None, want_dx=True,
           want_dres=False, gelu=False, p_drop=0.0, seed=0, out_scale=1.0, pen_coef=0.0, pen_coef_dev=None):
    _chk(x, BF16, "x"); _chk(dy, BF16, "dy"); _chk(dsum, BF16, "dsum"); _chk(aux, BF16, "aux")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    dx = empty(x.shape, x.dtype, x.device) if want_dx else None
    dres = empty(x.shape, x.dtype, x.device) if want_dres else None
    d = LnBwdDesc()
    d.x, d.gamma, d.beta, d.mean, d.rstd = _p(x), _p(gamma), _p(beta), _p(mean), _p(rstd)
    d.dy, d.dsum, d.aux, d.dx, d.dres, d.dgamma, d.dbeta = _p(dy), _p(dsum), _p(aux), _p(dx), _p(dres), _p(dgamma), _p(dbeta)
    d.rows, d.C, d.gelu, d.p_drop, d.seed, d.out_scale, d.pen_coef = rows, Cc, int(gelu), p_drop, seed, out_scale, pen_coef
    d.pen_coef_dev = _p(pen_coef_dev)
    ws = empty((768 * 2 * Cc,), torch.float32, x.device)     # dgamma/dbeta partial slab
    d.ws, d.ws_bytes = _p(ws), ws.numel() * 4
    _lib.call("w2vs_ln_bwd", C.byref(d), _stream())
    return dx, dres


# ------------------------------------------------------------------------------ encoder prologue
def _encpro_desc(x, mask, pad, pos, mask_emb, pos_table, gamma, beta, mean, rstd, src, p_in, seed_in, p_enc, seed_enc,
                 apply_ln, B, T, Tp, N, Cc):
    d = EncPrologueDesc()
    d.x, d.mask, d.pad, d.pos, d.mask_emb, d.pos_table = _p(x), _p(mask), _p(pad), _p(pos), _p(mask_emb), _p(pos_table)
    d.gamma, d.beta, d.mean, d.rstd, d.src = _p(gamma), _p(beta), _p(mean), _p(rstd), _p(src)
    d.p_in, d.p_enc, d.seed_in, d.seed_enc = p_in, p_enc, seed_in, seed_enc
    d.apply_ln, d.B, d.T, d.Tp, d.N, d.C = int(apply_ln), B, T, Tp, N, Cc
    return d


def enc_prologue_fwd(x, mask, pad, pos, mask_emb, pos_table, gamma, beta, src, Tp, *, apply_ln=True, p_in=0.0,
                     seed_in=0, p_enc=0.0, seed_enc=0):
    _chk(x, BF16, "x"); _chk(mask, torch.uint8, "mask"); _chk(pad, torch.uint8, "pad"); _chk(pos, torch.int32, "pos")
    _chk(pos_table, torch.float32, "pos_table"); _chk(src, torch.int32, "src")
    B, T, Cc = x.shape
    N = src.numel()
    out = empty((B, N, Cc), BF16, x.device)
    mean = empty((B * T), torch.float32, x.device)
    rstd = empty(mean.shape, mean.dtype, mean.device)
    d = _encpro_desc(x, mask, pad, pos, mask_emb, pos_table, gamma, beta, mean, rstd, src, p_in, seed_in, p_enc,
                     seed_enc, apply_ln, B, T, Tp, N, Cc)
    d.out = _p(out)
    _lib.call("w2vs_enc_prologue_fwd", C.byref(d), _stream())
    return out, mean, rstd


def enc_prologue_bwd(dout, x, mask, pad, pos, mask_emb, pos_table, gamma, beta, mean, rstd, src, copy_start, copy_list,
                     Tp, dmask_emb, dgamma, dbeta, *, apply_ln=True, p_in=0.0, seed_in=0, p_enc=0.0, seed_enc=0):
    _chk(dout, BF16, "dout")
    B, T, Cc = x.shape
    N = src.numel()
    dx = empty(x.shape, x.dtype, x.device)
    d = _encpro_desc(x, mask, pad, pos, mask_emb, pos_table, gamma, beta, mean, rstd, src, p_in, seed_in, p_enc,
                     seed_enc, apply_ln, B, T, Tp, N, Cc)
    d.dout, d.dx, d.dmask_emb, d.dgamma, d.dbeta = _p(dout), _p(dx), _p(dmask_emb), _p(dgamma), _p(dbeta)
    d.copy_start, d.copy_list = _p(copy_start), _p(copy_list)
    _lib.call("w2vs_enc_prologue_bwd", C.byref(d), _stream())
    return dx


# -------------------------------------------------------------------------------------- attention
def _attn_desc(qkv, o, lse, kpad, H, Tp, m, r, p_drop, seed):
    B, N, C3 = qkv.shape
    Cc = C3 // 3
    d = AttnDesc()
    base = qkv.data_ptr()
    d.q, d.k, d.v = C.c_void_p(base), C.c_void_p(base + 2 * Cc), C.c_void_p(base + 4 * Cc)
    d.o, d.lse, d.kpad = _p(o), _p(lse), _p(kpad)
    d.ld, d.ldo, d.sb, d.sbo = C3, Cc, N * C3, N * Cc
    d.B, d.H, d.N, d.Tp, d.m, d.r, d.head_dim = B, H, N, Tp, m, r, Cc // H
    d.scale, d.p_drop, d.seed = float(Cc // H) ** -0.5, p_drop, seed
    return d


def attn_tune(variant=-1):
    """1 = attention.hip (128-query workgroups), 2 = attention2.hip (32-row workgroups, waves split the long dimension),
    -1 = default."""
    _lib.call("w2vs_attn_tune", variant)


def attn_drop_bits(B, H, N, Nq=0, device="cuda"):
    """Scratch for the attention-dropout keep masks (w2vs_attn_desc.drop_bits): pass the same tensor to attn_fwd and attn_bwd."""
    lib = _lib.load()
    lib.w2vs_attn_drop_bits_bytes.restype = C.c_int64
    return empty((int(lib.w2vs_attn_drop_bits_bytes(B, H, N, Nq)) // 4,), torch.int32, device)


def attn_fwd(qkv, H, Tp, m, r, kpad=None, p_drop=0.0, seed=0, drop_bits=None):
    """qkv [B, N, 3C] bf16 (q | k | v).  Returns ctx [B, N, C] and lse [B, H, N]."""
    _chk(qkv, BF16, "qkv"); _chk(kpad, torch.uint8, "kpad"); _chk(drop_bits, torch.int32, "drop_bits")
    B, N, C3 = qkv.shape
    o = empty((B, N, C3 // 3), BF16, qkv.device)
    lse = empty((B, H, N), torch.float32, qkv.device)
    d = _attn_desc(qkv, o, lse, kpad, H, Tp, m, r, p_drop, seed)
    d.drop_bits = _p(drop_bits)
    _lib.call("w2vs_attn_fwd", C.byref(d), _stream())
    return o, lse


def attn_bwd(dout, qkv, o, lse, H, Tp, m, r, kpad=None, p_drop=0.0, seed=0, drop_bits=None):
    _chk(dout, BF16, "dout"); _chk(qkv, BF16, "qkv"); _chk(o, BF16, "o")
    B, N, C3 = qkv.shape
    Cc = C3 // 3
    dqkv = empty(qkv.shape, qkv.dtype, qkv.device)
    delta = empty((B, H, N), torch.float32, qkv.device)
    d = _attn_desc(qkv, o, lse, kpad, H, Tp, m, r, p_drop, seed)
    base = dqkv.data_ptr()
    d.dout, d.delta = _p(dout), _p(delta)
    d.dq, d.dk, d.dv = C.c_void_p(base), C.c_void_p(base + 2 * Cc), C.c_void_p(base + 4 * Cc)
    d.drop_bits = _p(drop_bits)
    _lib.call("w2vs_attn_bwd", C.byref(d), _stream())
    return dqkv


def _cross_desc(q, kv, o, lse, kpad, H, m, U, p_drop, seed):
    B, Nq, Cq = q.shape
    _, S, C2 = kv.shape
    Cc = C2 // 2
    d = AttnDesc()
    base = kv.data_ptr()
    d.q, d.k, d.v = _p(q), C.c_void_p(base), C.c_void_p(base + 2 * Cc)
    d.o, d.lse, d.kpad = _p(o), _p(lse), _p(kpad)
    d.ld, d.ldo, d.sb, d.sbo = C2, Cc, S * C2, Nq * Cc
    d.B, d.H, d.N, d.Tp, d.m, d.r, d.head_dim = B, H, S, S, m, 0, Cc // H
    d.scale, d.p_drop, d.seed = float(Cc // H) ** -0.5, p_drop, seed
    d.Nq, d.mq, d.ldq, d.sbq = Nq, U, Cq, Nq * Cq
    return d


def group_attn_fwd(q, kv, H, m, U, kpad=None, p_drop=0.0, seed=0):
    """Group-prefix cross attention (w2vs_attn_desc cross mode): q [B, G*U, C] rows (g, u); kv [B, S, 2C] (k | v); query
    row (g, u) attends the keys < min((g + 1) * m, S) not marked in kpad [B, S].  Returns ctx [B, G*U, C], lse [B, H, G*U]."""
    _chk(q, BF16, "q"); _chk(kv, BF16, "kv"); _chk(kpad, torch.uint8, "kpad")
    B, Nq, Cc = q.shape
    o = empty((B, Nq, Cc), BF16, q.device)
    lse = empty((B, H, Nq), torch.float32, q.device)
    d = _cross_desc(q, kv, o, lse, kpad, H, m, U, p_drop, seed)
    _lib.call("w2vs_attn_fwd", C.byref(d), _stream())
    return o, lse


def group_attn_bwd(dout, q, kv, o, lse, H, m, U, kpad=None, p_drop=0.0, seed=0):
    """Returns dq [B, G*U, C] and dkv [B, S, 2C]."""
    _chk(dout, BF16, "dout"); _chk(q, BF16, "q"); _chk(kv, BF16, "kv"); _chk(o, BF16, "o")
    B, Nq, Cc = q.shape
    dq = empty(q.shape, BF16, q.device)
    dkv = empty(kv.shape, BF16, q.device)
    delta = empty((B, H, Nq), torch.float32, q.device)
    d = _cross_desc(q, kv, o, lse, kpad, H, m, U, p_drop, seed)
    base = dkv.data_ptr()
    d.dout, d.delta, d.dq = _p(dout), _p(delta), _p(dq)
    d.dk, d.dv = C.c_void_p(base), C.c_void_p(base + 2 * Cc)
    _lib.call("w2vs_attn_bwd", C.byref(d), _stream())
    return dq, dkv


# -------------------------------------------------------------------------------------- quantizer
class QuantState:
    __slots__ = ("idx", "hard_cnt", "prob_sum", "ppl", "cvec")


def _quant_logits(d, logits, bias):
    if logits.dtype == torch.float32:
        _chk(logits, torch.float32, "logits"); _chk(bias, BF16, "logit bias")
        d.logits_f32, d.logit_bias = _p(logits), _p(bias)
    else:
        _chk(logits, BF16, "logits")
        if bias is not None:
            raise W2vsError("bf16 quantizer logits already carry their bias")
        d.logits = _p(logits)


def linear_fwd_f32(x, w):
    """x @ w.T as fp32, no bias (W2VS_EPI_F32): the quantizer logits, whose argmax must not see bf16 rounding."""
    _chk(x, BF16, "x"); _chk(w, BF16, "w")
    R, K = x.shape
    N = w.shape[0]
    y = empty((R, N), torch.float32, x.device)
    gemm_nt(x, w, M=R, N=N, K=K, lda=K, ldb=K, ldc=N, out_f32=y, epi=EPI_F32)
    return y


def quant_fwd(logits, vars2d, G, V, tau, training, noise=None, seed=0, bias=None):
    """logits [R, G*V] (fp32 without bias + ``bias`` [G*V] bf16, or bf16 with the bias folded in), vars2d [G*V, D] bf16
    -> q [R, G*D], state (ppl = [prob_ppl, code_ppl])."""
    _chk(vars2d, BF16, "vars"); _chk(noise, torch.float32, "noise")
    R = logits.shape[0]
    D = vars2d.shape[1]
    dev = logits.device
    q = empty((R, G * D), BF16, dev)
    st = QuantState()
    st.idx = empty((R, G), torch.int32, dev)
    st.hard_cnt = empty((G * V), torch.float32, dev)
    st.prob_sum = empty((G * V), torch.float32, dev)
    st.ppl = empty((2), torch.float32, dev)
    st.cvec = empty((G * V), torch.float32, dev)
    d = QuantDesc()
    _quant_logits(d, logits, bias)
    d.noise, d.vars, d.q, d.idx = _p(noise), _p(vars2d), _p(q), _p(st.idx)
    d.hard_cnt, d.prob_sum, d.ppl_out, d.cvec_out = _p(st.hard_cnt), _p(st.prob_sum), _p(st.ppl), _p(st.cvec)
    d.tau, d.R, d.G, d.V, d.D, d.training, d.seed = tau, R, G, V, D, int(training), seed
    _lib.call("w2vs_quant_fwd", C.byref(d), _stream())
    return q, st


def quant_bwd(dq, logits, vars2d, st, G, V, tau, training, ppl_grad, dvars_f32, noise=None, seed=0, ppl_grad_dev=None,
              bias=None):
    """Returns dlogits [R, G*V] bf16; accumulates dvars (fp32 [G*V, D])."""
    _chk(dq, BF16, "dq")
    R = logits.shape[0]
    D = vars2d.shape[1]
    dsoft = None
    if training:
        dsoft = empty((R, G * V), torch.float32, dq.device)
        # dsoft[:, g] = dq[:, g] @ vars_g^T : batched over groups through column-offset strides; fp32 out: the softmax
        # backward subtracts <soft, dsoft> from it, and a bf16 dsoft loses ~5 % of weight_proj's gradient to that cancellation
        gemm_nt(dq, vars2d, M=R, N=V, K=D, lda=G * D, ldb=D, ldc=G * V, out_f32=dsoft, epi=EPI_F32, batch=G, sA=D, sB=V * D, sC=V,
                a_bytes=(R * G * D) * 2, b_bytes=V * D * 2, c_elems=R * G * V)
    dlogits = empty((R, G * V), BF16, dq.device)
    d = QuantDesc()
    _quant_logits(d, logits, bias)
    d.noise, d.vars = _p(noise), _p(vars2d)
    d.prob_sum, d.dq, d.dsoft_f32, d.cvec, d.dlogits, d.dvars = _p(st.prob_sum), _p(dq), _p(dsoft), _p(st.cvec), _p(dlogits), _p(dvars_f32)
    d.ppl_grad, d.tau, d.R, d.G, d.V, d.D, d.training, d.seed = ppl_grad, tau, R, G, V, D, int(training), seed
    d.ppl_grad_dev = _p(ppl_grad_dev)
    _lib.call("w2vs_quant_bwd", C.byref(d), _stream())
    return dlogits


# ---------------------------------------------------------------------------------------- InfoNCE
def nce_fwd(x, y, neg_idx, B, M, K, temp):
    """x, y [B*M, C] bf16; neg_idx [B, K*M] int64 -> logits [B*M, K+1] fp32 and the saved row norms."""
    _chk(x, BF16, "x"); _chk(y, BF16, "y"); _chk(neg_idx, torch.int64, "neg_idx")
    Cc = x.shape[1]
    logits = empty((B * M, K + 1), torch.float32, x.device)
    norms = empty((2, B * M), torch.float32, x.device)
    d = NceDesc()
    d.x, d.y, d.neg_idx, d.logits = _p(x), _p(y), _p(neg_idx), _p(logits)
    d.xn, d.yn = _p(norms[0]), _p(norms[1])
    d.B, d.M, d.K, d.C, d.temp = B, M, K, Cc, temp
    _lib.call("w2vs_nce_fwd", C.byref(d), _stream())
    return logits, norms


def nce_bwd(dlogits, logits, norms, x, y, neg_idx, B, M, K, temp):
    """Returns dx, dy bf16 [B*M, C]."""
    _chk(dlogits, torch.float32, "dlogits"); _chk(logits, torch.float32, "logits")
    Cc = x.shape[1]
    dx = empty((B * M, Cc), BF16, x.device)
    dy = empty((B * M, Cc), BF16, x.device)
    d = NceDesc()
    d.x, d.y, d.neg_idx, d.logits, d.dlogits, d.dx, d.dy = _p(x), _p(y), _p(neg_idx), _p(logits), _p(dlogits), _p(dx), _p(dy)
    ws = empty((B * M, Cc), torch.float32, x.device)
    d.dy_ws = _p(ws)
    Mp = (M + 63) // 64 * 64
    ws2 = empty((B * Mp * (Mp + 2 * Cc + 2),), torch.float32, x.device)
    d.ws, d.ws_bytes = _p(ws2), ws2.numel() * 4
    d.xn, d.yn = _p(norms[0]), _p(norms[1])
    d.B, d.M, d.K, d.C, d.temp = B, M, K, Cc, temp
    _lib.call("w2vs_nce_bwd", C.byref(d), _stream())
    return dx, dy


def ce_rows(logits, want_grad=True):
    """Cross entropy (target 0, sum) of fp32 logits [R, W].  Returns out3 = [loss, n_max0, n_both0], dlogits."""
    _chk(logits, torch.float32, "logits")
    R, W = logits.shape
    out3 = empty((3), torch.float32, logits.device)
    dl = empty(logits.shape, logits.dtype, logits.device) if want_grad else None
    _lib.call("w2vs_ce_rows", _p(logits), R, W, _p(out3), _p(dl), _stream())
    return out3, dl


_LOSS_SCRATCH = {}


def infonce_loss(logits, pen_acc, ppl, *, w_ppl, w_pen, num_vars, pen_norm, sample_size, want_grad=True):
    """The InfoNCE criterion's arithmetic in ONE launch (w2vs_infonce_loss): cross entropy of the fp32 logits [R, W] against
    class 0, then loss = ce + w_ppl * ((num_vars - prob_ppl) / num_vars) * sample_size + w_pen * pen_acc * pen_norm * sample_size.
    Returns (loss [1], vec [8] = loss, ce, ppl term, pen term, correct, prob_ppl, code_ppl, features_pen, dlogits or None).
    loss and vec come from torch's allocator, not the step arena: they are handed to autograd / the caller's logging."""
    _chk(logits, torch.float32, "logits"); _chk(pen_acc, torch.float32, "pen_acc"); _chk(ppl, torch.float32, "ppl")
    R, W = logits.shape
    dev = logits.device
    scratch = _LOSS_SCRATCH.get(dev)
    if scratch is None:
        scratch = _LOSS_SCRATCH[dev] = torch.zeros(4, dtype=torch.float32, device=dev)   # the kernel leaves it zero
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    vec = torch.empty(8, dtype=torch.float32, device=dev)
    dl = empty(logits.shape, logits.dtype, dev) if want_grad else None
    d = InfonceLossDesc()
    d.logits, d.R, d.W, d.pen_acc, d.ppl = _p(logits), R, W, _p(pen_acc), _p(ppl)
    d.w_ppl, d.w_pen, d.num_vars, d.pen_norm, d.sample_size = w_ppl, w_pen, num_vars, pen_norm, sample_size
    d.loss, d.vec, d.dlogits, d.scratch = _p(loss), _p(vec), _p(dl), _p(scratch)
    _lib.call("w2vs_infonce_loss", C.byref(d), _stream())
    return loss, vec, dl


def infonce_loss_bwd(g, dlogits, c_pen, c_ppl):
    """dlogits *= g in place; returns dsc [2] = (g * c_pen, g * c_ppl): the gradients of features_pen and prob_perplexity."""
    _chk(g, torch.float32, "g"); _chk(dlogits, torch.float32, "dlogits")
    dsc = empty((2,), torch.float32, dlogits.device)
    _lib.call("w2vs_infonce_loss_bwd", _p(g), _p(dlogits), dlogits.numel(), c_pen, c_ppl, _p(dsc), _stream())
    return dsc


def dropout(x, p, seed):
    _chk(x, BF16, "x")
    out = empty(x.shape, x.dtype, x.device)
    _lib.call("w2vs_dropout", _p(x), _p(out), x.numel(), p, seed, _stream())
    return out


def relu_gate(x, gate):
    """gate > 0 ? x : 0 (bf16).  relu(x) = relu_gate(x, x); its backward = relu_gate(dy, y)."""
    _chk(x, BF16, "x"); _chk(gate, BF16, "gate")
    out = empty(x.shape, x.dtype, x.device)
    _lib.call("w2vs_relu_gate", _p(x), _p(gate), _p(out), x.numel(), _stream())
    return out


def adam_step(p32, p16, m, v, g, *, lr, beta1, beta2, eps, weight_decay, step, scale_host=1.0, scale_dev=None):
    _chk(p32, torch.float32, "p32"); _chk(p16, BF16, "p16"); _chk(g, torch.float32, "g")
    _lib.call("w2vs_adam_step", _p(p32), _p(p16), _p(m), _p(v), _p(g), p32.numel(), lr, beta1, beta2, eps, weight_decay,
              step, _p(scale_dev), scale_host, _stream())


def sumsq(x, out):
    _chk(x, torch.float32, "x"); _chk(out, torch.float32, "out")
    _lib.call("w2vs_sumsq", _p(x), x.numel(), _p(out), _stream())


def clip_scale(sumsq_buf, out3, *, scale_host=1.0, scale_dev=None, clip=0.0):
    """out3 = [grad scale incl. the clip factor, gnorm, non-finite flag], all on the device (w2vs_clip_scale)."""
    _chk(sumsq_buf, torch.float32, "sumsq"); _chk(out3, torch.float32, "out3"); _chk(scale_dev, torch.float32, "scale_dev")
    _lib.call("w2vs_clip_scale", _p(sumsq_buf), _p(scale_dev), scale_host, clip, _p(out3), _stream())


def clip_scale_acc(sumsq_buf, out3, bad_acc, *, scale_host=1.0, scale_dev=None, clip=0.0):
    """clip_scale that also clears ``sumsq_buf`` for the next update and adds the non-finite flag to ``bad_acc``
    (w2vs_clip_scale_acc): one launch instead of three at the end of every update."""
    _chk(sumsq_buf, torch.float32, "sumsq"); _chk(out3, torch.float32, "out3"); _chk(scale_dev, torch.float32, "scale_dev")
    _chk(bad_acc, torch.float32, "bad_acc")
    _lib.call("w2vs_clip_scale_acc", _p(sumsq_buf), _p(scale_dev), scale_host, clip, _p(out3), _p(bad_acc), _stream())


def gather_rows(src, idx, R, scatter=False, out=None):
    """out[i] = src[idx[i]]  or (scatter) out[idx[i]] = src[i]; rows of bf16."""
    _chk(src, BF16, "src"); _chk(idx, torch.int32, "idx")
    Cc = src.shape[-1]
    if out is None:
        if scatter:
            raise W2vsError("scatter needs a pre-zeroed destination")
        out = empty((R, Cc), BF16, src.device)
    _lib.call("w2vs_gather_rows", _p(src), _p(idx), _p(out), R, Cc, int(scatter), _stream())
    return out
