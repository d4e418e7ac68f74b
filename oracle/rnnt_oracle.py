"""CPU oracle for SURVEY.md section 8 row f4: the RNN-T / delay-transducer loss of
``/root/reference/warp_transducer``.  TEST INFRASTRUCTURE ONLY (same rules as w2vs_oracle.py:
only tests/, smoke() and bench.py's cpu_baseline leg may import this).

A float64 numpy restatement of the reference's GPU path, one function per kernel, each citing
``warp_transducer/include/detail/`` ("wt/"):

  log_softmax_denom   reduce_max + reduce_exp            wt/reduce.h:46-123, delay_transducer.h:84-90
  alphas / betas      compute_alphas/betas_kernel        wt/gpu_rnnt_kernel.h:12-51, 126-163
  alpha_delay / beta_delay                               wt/gpu_rnnt_kernel.h:54-100, 166-213
  grads               compute_grad_kernel (:248-285) and compute_grad_withdelay_smooth_kernel (:373-426)
  delay_loss          DelayTransducer::compute_cost_and_score, wt/delay_transducer.h:92-380

Pinning.  The plain transducer part (costs, alphas/betas, gradients) is pinned by
  * the reference's own CPU implementation compiled from its sources (``make -C oracle ref`` ->
    ``oracle/_ref/libwarprnnt_cpu.so`` from wt/../src/rnnt_entrypoint.cpp + wt/cpu_rnnt.h), driven through
    ``RefCpuRnnt`` below, and
  * the known answers the reference's tests hold (tests/golden/rnnt_known_answers.json, transcribed from
    warp_transducer/tests/test_gpu.cu:17-222, test_delay.cu:33-176, test_cpu.cpp:14-130).
The delay part (alpha_delay, beta_delay, expected delay, delay gradient) exists ONLY as CUDA in the reference
and its tests for it are stale (test_delay.cu still builds B x T delay values for kernels that index
B x T x U, and runs grad_check with delay_scale = 0), so no reference output can be produced for it here:
PARITY UNPINNED for the delay terms.  They are anchored instead on invariants the algorithm implies
(expected delay from the forward and the backward recursion agree; the occupancy-weighted cell costs
sum to the same expectation on every anti-diagonal; with ``consistent_delay_index`` the analytic gradient
equals a central-difference gradient of cost_rnnt + delay_scale * cost_delay).

The loss head (``transducer_out_step`` / ``label_smoothed_ce``, rain/layers/attention_transducer.py:289-408 and
fs/criterions/label_smoothed_cross_entropy.py:33-50) is a restatement too and likewise PARITY UNPINNED: that
reference file imports the CUDA-only ``warprnnt_pytorch`` and the fairseq Transformer stack, so it cannot be run here;
tests/test_rnnt_oracle_cpu.py checks the restatement against torch autograd through the same composition.

Reference quirk kept on purpose: the gradient kernels read ``delay_values[bt]`` (bt = b * maxT + t,
:409) from the B x T x U array that the alpha/beta kernels index as ``[b, t, u]`` (:76, :186).  The
restatement does the same by default (``consistent_delay_index=False``); ``True`` uses ``[b, t, u]``.
"""
import ctypes as C
import os

import numpy as np

NEG_INF = -np.inf


def log_softmax_denom(acts):
    """denom[b,t,u] = -max - log(sum(exp(acts - max))), so log p = denom + acts."""
    m = acts.max(-1)
    return -m - np.log(np.exp(acts - m[..., None]).sum(-1))


def _lse(a, b):
    if a == NEG_INF:
        return b
    if b == NEG_INF:
        return a
    return np.log1p(np.exp(-abs(a - b))) + max(a, b)


def alphas(lp, labels, T, U, blank):
    """lp [maxT, maxU, V] log-probs of one sample -> (alphas [T, U], log-likelihood)."""
    a = np.zeros((T, U))
    for t in range(T):
        for u in range(U):
            if u == 0 and t > 0:
                a[t, 0] = a[t - 1, 0] + lp[t - 1, 0, blank]
            if t == 0 and u > 0:
                a[0, u] = a[0, u - 1] + lp[0, u - 1, labels[u - 1]]
            if t > 0 and u > 0:
                a[t, u] = _lse(a[t, u - 1] + lp[t, u - 1, labels[u - 1]], a[t - 1, u] + lp[t - 1, u, blank])
    return a, a[T - 1, U - 1] + lp[T - 1, U - 1, blank]


def betas(lp, labels, T, U, blank):
    b = np.zeros((T, U))
    b[T - 1, U - 1] = lp[T - 1, U - 1, blank]
    for t in range(T - 1, -1, -1):
        for u in range(U - 1, -1, -1):
            if u == U - 1 and t < T - 1:
                b[t, u] = b[t + 1, u] + lp[t, u, blank]
            if t == T - 1 and u < U - 1:
                b[t, u] = b[t, u + 1] + lp[t, u, labels[u]]
            if t < T - 1 and u < U - 1:
                b[t, u] = _lse(b[t, u + 1] + lp[t, u, labels[u]], b[t + 1, u] + lp[t, u, blank])
    return b, b[0, 0]


def alpha_delay(lp, a, dv, labels, T, U, blank):
    """Expected accumulated delay of the paths reaching (t, u): emitting label u at frame t costs dv[t, u]."""
    ad = np.zeros((T, U))
    for t in range(T):
        for u in range(1, U):
            if t == 0:
                ad[0, u] = ad[0, u - 1] + dv[0, u]
            else:
                no_emit = np.exp(a[t - 1, u] + lp[t - 1, u, blank] - a[t, u]) * ad[t - 1, u]
                emit = np.exp(a[t, u - 1] + lp[t, u - 1, labels[u - 1]] - a[t, u]) * (ad[t, u - 1] + dv[t, u])
                ad[t, u] = no_emit + emit
    return ad, ad[T - 1, U - 1]


def beta_delay(lp, b, dv, labels, T, U, blank):
    bd = np.zeros((T, U))
    for t in range(T - 1, -1, -1):
        for u in range(U - 1, -1, -1):
            if u == U - 1:
                if t < T - 1:
                    bd[t, u] = bd[t + 1, u]
            elif t == T - 1:
                bd[t, u] = bd[t, u + 1] + dv[t, u]
            else:
                no_emit = np.exp(b[t + 1, u] + lp[t, u, blank] - b[t, u]) * bd[t + 1, u]
                emit = np.exp(b[t, u + 1] + lp[t, u, labels[u]] - b[t, u]) * (bd[t, u + 1] + dv[t, u])
                bd[t, u] = no_emit + emit
    return bd, bd[0, 0]


def rnnt_loss(acts, labels, input_lengths, label_lengths, blank=0, want_grad=True):
    """compute_rnnt_loss with loc = GPU (wt/gpu_rnnt.h:78-215): raw activations [B, T, U, V] in, costs [B] and
    gradients w.r.t. the ACTIVATIONS out (zeros outside each sample's T x U)."""
    acts = np.asarray(acts, dtype=np.float64)
    B, maxT, maxU, V = acts.shape
    labels = np.asarray(labels).reshape(B, maxU - 1)
    denom = log_softmax_denom(acts)
    lp = acts + denom[..., None]
    costs = np.zeros(B)
    grads = np.zeros_like(acts) if want_grad else None
    for mb in range(B):
        T, U = int(input_lengths[mb]), int(label_lengths[mb]) + 1
        a, ll = alphas(lp[mb], labels[mb], T, U, blank)
        costs[mb] = -ll
        if not want_grad:
            continue
        b, _ = betas(lp[mb], labels[mb], T, U, blank)
        for t in range(T):
            for u in range(U):
                g = np.exp(a[t, u] + b[t, u] + lp[mb, t, u] - ll)
                if t == T - 1 and u == U - 1:
                    g[blank] -= np.exp(a[t, u] + lp[mb, t, u, blank] - ll)
                if t < T - 1:
                    g[blank] -= np.exp(a[t, u] + lp[mb, t, u, blank] - ll + b[t + 1, u])
                if u < U - 1:
                    l = labels[mb, u]
                    g[l] -= np.exp(a[t, u] + lp[mb, t, u, l] - ll + b[t, u + 1])
                grads[mb, t, u] = g
    return costs, grads


def delay_loss(acts, labels, input_lengths, label_lengths, delay_values, delay_scale=1.0, smooth=1.0, blank=0,
               want_grad=True, consistent_delay_index=False, collect=None):
    """compute_rnnt_delay_loss (wt/delay_transducer.h:92-380).  Returns (costs [3, B] = NLL, expected delay,
    NLL + delay_scale * expected delay; gradients [B, T, U, V] or None)."""
    acts = np.asarray(acts, dtype=np.float64)
    B, maxT, maxU, V = acts.shape
    labels = np.asarray(labels).reshape(B, maxU - 1)
    dv_all = np.asarray(delay_values, dtype=np.float64).reshape(B, maxT, maxU)
    dv_flat = dv_all.reshape(-1)
    denom = log_softmax_denom(acts)
    lp = acts + denom[..., None]
    costs = np.zeros((3, B))
    grads = np.zeros_like(acts) if want_grad else None
    for mb in range(B):
        T, U = int(input_lengths[mb]), int(label_lengths[mb]) + 1
        lab, dv = labels[mb], dv_all[mb]
        a, ll = alphas(lp[mb], lab, T, U, blank)
        ad, dexp = alpha_delay(lp[mb], a, dv, lab, T, U, blank)
        costs[0, mb] = -ll
        costs[1, mb] = dexp
        costs[2, mb] = -ll + delay_scale * dexp
        if collect is not None:
            collect.setdefault("alphas", []).append(a)
            collect.setdefault("alpha_delay", []).append(ad)
        if not want_grad and collect is None:
            continue
        b, _ = betas(lp[mb], lab, T, U, blank)
        bd, dexp_b = beta_delay(lp[mb], b, dv, lab, T, U, blank)
        if collect is not None:
            collect.setdefault("betas", []).append(b)
            collect.setdefault("beta_delay", []).append(bd)
            collect.setdefault("delay_expect_bwd", []).append(dexp_b)
        if not want_grad:
            continue
        for t in range(T):
            for u in range(U):
                logpk = lp[mb, t, u]
                logpb = logpk[blank]
                logpy = logpk[lab[u]] if u < U - 1 else 0.0
                g = np.exp((a[t, u] + b[t, u] - ll) * smooth + logpk)
                g2 = np.zeros(V)
                c0 = c1 = 0.0
                if t < T - 1:
                    c0 = ad[t, u] + bd[t + 1, u] - dexp
                    g2 -= np.exp(a[t, u] + b[t + 1, u] + logpk - ll + logpb) * c0
                if u < U - 1:
                    d = dv[t, u] if consistent_delay_index else dv_flat[mb * maxT + t]     # :409 reads [bt]
                    c1 = ad[t, u] + d + bd[t, u + 1] - dexp
                    g2 -= np.exp(a[t, u] + b[t, u + 1] + logpk - ll + logpy) * c1
                if t == T - 1 and u == U - 1:
                    g[blank] -= np.exp(smooth * (a[t, u] - ll + logpb))
                if t < T - 1:
                    g[blank] -= np.exp(smooth * (a[t, u] - ll + b[t + 1, u] + logpb))
                    g2[blank] += np.exp(a[t, u] + b[t + 1, u] + logpb - ll) * c0
                if u < U - 1:
                    l = lab[u]
                    g[l] -= np.exp(smooth * (a[t, u] + b[t, u + 1] - ll + logpk[l]))
                    g2[l] += np.exp(a[t, u] + b[t, u + 1] + logpy - ll) * c1
                grads[mb, t, u] = g + delay_scale * g2
    return costs, grads


# delay-cost builders of the Python front end, warp_transducer/pytorch_binding/warprnnt_pytorch/delay_transducer.py:96-134
def delay_cost(kind, B, T, U, src_lens, tgt_lens):
    src_lens = np.asarray(src_lens, dtype=np.float64)
    tgt_lens = np.asarray(tgt_lens, dtype=np.float64)
    s = np.arange(T, dtype=np.float64)[None, :, None]
    u = np.arange(U, dtype=np.float64)[None, None, :]
    if kind == "zero":
        return np.broadcast_to(s / src_lens[:, None, None], (B, T, U)).copy()
    gamma = (tgt_lens / src_lens)[:, None, None]
    d = (s + 1) * gamma - (u + 1)
    d = np.abs(d) if kind == "diagonal" else np.clip(d, 0, None)
    assert kind in ("diagonal", "diag_positive"), kind
    return d / tgt_lens[:, None, None]


# ----------------------------------------------------------------------------------------------------
# the reference's own CPU implementation, compiled by `make -C oracle ref` (checker for the plain RNN-T part)
# ----------------------------------------------------------------------------------------------------
class _RnntOptions(C.Structure):
    _fields_ = [("loc", C.c_int), ("num_threads", C.c_uint), ("stream", C.c_void_p), ("blank_label", C.c_int),
                ("maxT", C.c_int), ("maxU", C.c_int), ("batch_first", C.c_bool)]


class RefCpuRnnt:
    """ctypes driver of oracle/_ref/libwarprnnt_cpu.so (CpuRNNT, wt/cpu_rnnt.h).  That implementation takes
    LOG-PROBABILITIES and returns gradients w.r.t. them (warprnnt_pytorch/rnnt.py:67-68 applies log_softmax
    first); ``loss_and_act_grads`` chains through the softmax so the result is comparable with the GPU path."""
    PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libwarprnnt_cpu.so")

    @classmethod
    def available(cls):
        return os.path.isfile(cls.PATH)

    def __init__(self):
        self.lib = C.CDLL(self.PATH)
        self.lib.compute_rnnt_loss.restype = C.c_int
        self.lib.get_workspace_size.restype = C.c_int

    def loss_and_logprob_grads(self, log_probs, labels, input_lengths, label_lengths, blank=0):
        lp = np.ascontiguousarray(log_probs, dtype=np.float32)
        B, T, U, V = lp.shape
        grads = np.zeros_like(lp)
        costs = np.zeros(B, dtype=np.float32)
        lab = np.ascontiguousarray(labels, dtype=np.int32)
        xl = np.ascontiguousarray(input_lengths, dtype=np.int32)
        yl = np.ascontiguousarray(label_lengths, dtype=np.int32)
        size = C.c_size_t(0)
        assert self.lib.get_workspace_size(T, U, B, C.c_bool(False), C.byref(size), C.c_size_t(4)) == 0
        ws = np.zeros(size.value + 64, dtype=np.uint8)
        opt = _RnntOptions(0, 1, None, blank, T, U, True)
        rc = self.lib.compute_rnnt_loss(lp.ctypes.data_as(C.c_void_p), grads.ctypes.data_as(C.c_void_p),
                                        lab.ctypes.data_as(C.c_void_p), yl.ctypes.data_as(C.c_void_p),
                                        xl.ctypes.data_as(C.c_void_p), V, B, costs.ctypes.data_as(C.c_void_p),
                                        ws.ctypes.data_as(C.c_void_p), opt)
        assert rc == 0, rc
        return costs, grads

    def loss_and_act_grads(self, acts, labels, input_lengths, label_lengths, blank=0):
        acts = np.asarray(acts, dtype=np.float64)
        lp = acts + log_softmax_denom(acts)[..., None]
        costs, g = self.loss_and_logprob_grads(lp, labels, input_lengths, label_lengths, blank)
        g = g.astype(np.float64)
        return costs, g - np.exp(lp) * g.sum(-1, keepdims=True)


# ----------------------------------------------------------------------------------------------------
# the CAAT loss head: TransducerOut.train_step, rain/layers/attention_transducer.py:289-408
# ----------------------------------------------------------------------------------------------------
def label_smoothed_ce(logits, target, eps, pad):
    """fs/criterions/label_smoothed_cross_entropy.py:33-50 on log_softmax(logits), summed, rows with target == pad
    ignored.  Returns (loss, nll, d loss / d logits)."""
    logits = np.asarray(logits, dtype=np.float64)
    V = logits.shape[-1]
    lp = logits + log_softmax_denom(logits)[..., None]
    keep = (target != pad)
    rows = np.arange(logits.shape[0])
    nll_r = -lp[rows, np.where(keep, target, 0)] * keep
    smooth_r = -lp.sum(-1) * keep
    eps_i = eps / (V - 1)
    loss = (1.0 - eps - eps_i) * nll_r.sum() + eps_i * smooth_r.sum()
    p = np.exp(lp)
    g = (1.0 - eps - eps_i) * p + eps_i * (V * p - 1.0)
    g[rows, np.where(keep, target, 0)] -= (1.0 - eps - eps_i)
    g *= keep[:, None]
    return loss, nll_r.sum(), g


def transducer_out_step(x, W, targets, src_lengths, tgt_lengths, *, delay_scale=1.0, temperature=1.0, blank=0,
                        label_smoothing=0.1, pad=1, ce_scale=1.0, delay_func="zero", loss_scale=1.0, tokens_per_step=None):
    """TransducerOut.train_step (:362-408).  x [B, T, U, d] joint states, W [V, d] bias-free output projection, targets
    [B, U-1].  Returns the result dictionary's four losses and (d x, d W) of loss_scale * (rnnt_total + ce_scale * ce).
    ``tokens_per_step`` splits the batch exactly as :370-373 does.  Every term is a sum over samples, so the split would
    not matter - except that the gradient kernel's ``delay_values[b * maxT + t]`` index (see module docstring) uses the
    micro-batch-local b: with the reference's index the delay gradient DOES depend on the micro-batching."""
    x = np.asarray(x, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    B, T, U, d = x.shape
    src_lengths, tgt_lengths, targets = np.asarray(src_lengths), np.asarray(tgt_lengths), np.asarray(targets)
    step = B if tokens_per_step is None else max(tokens_per_step // (T * U), 1)
    out = {"loss": 0.0, "loss_prob": 0.0, "loss_delay": 0.0, "nll_loss": 0.0}
    dx = np.zeros_like(x)
    dW = np.zeros_like(W)
    for i in range(0, B, step):
        sl = slice(i, min(i + step, B))
        xb, n = x[sl], x[sl].shape[0]
        logits = xb @ W.T
        dv = delay_cost(delay_func, n, T, U, src_lengths[sl], tgt_lengths[sl])
        costs, dl = delay_loss(logits, targets[sl], src_lengths[sl], tgt_lengths[sl], dv, delay_scale=delay_scale,
                               smooth=temperature, blank=blank)
        dl = dl * loss_scale
        dxb = dl @ W
        dW += dl.reshape(-1, W.shape[0]).T @ xb.reshape(-1, d)
        bidx = np.arange(n)
        last_h = xb[bidx, src_lengths[sl] - 1][:, :-1]                              # :345-348
        ce, nll, g2 = label_smoothed_ce((last_h @ W.T).reshape(n * (U - 1), -1), targets[sl].reshape(-1), label_smoothing, pad)
        g2 = g2.reshape(n, U - 1, -1) * (ce_scale * loss_scale)
        dxb[bidx, src_lengths[sl] - 1, :U - 1] += g2 @ W
        dW += g2.reshape(-1, W.shape[0]).T @ last_h.reshape(-1, d)
        dx[sl] = dxb
        out["loss"] += costs[2].sum() + ce_scale * ce
        out["loss_prob"] += costs[0].sum()
        out["loss_delay"] += costs[1].sum()
        out["nll_loss"] += nll
    return out, dx, dW


# ------------------------------------------------------------------------------------------------------------------
# CAAT joint network (rain/layers/attention_transducer.py:591-852), fp32 torch restatement.  Test infrastructure only.
# ------------------------------------------------------------------------------------------------------------------
def mha_joint_net(P, enc, pad, dec, *, layers, heads, downsample, normalize_before=True, prefix="layers."):
    """MHAJointNet.forward (:826-852) with eval-mode dropouts.  P: name -> fp32 tensor (reference state_dict names),
    enc [S, B, D] encoder frames, pad [B, S] bool, dec [B, U, D] decoder states.
    Returns x [B, G, U, D] and group_lengths [B] (G = ceil(S / downsample); downsample <= 0: one group, no mask)."""
    import math as _m
    import torch
    import torch.nn.functional as F
    S, B, D = enc.shape
    U = dec.shape[1]
    hd = D // heads
    if downsample > 0:                                  # _gen_group_mask (:810-824)
        G = _m.ceil(S / downsample)
        gpos = torch.arange(1, G + 1) * downsample
        gmask = torch.zeros(G, S).masked_fill(gpos.unsqueeze(1) <= torch.arange(S).unsqueeze(0), float("-inf"))
        glen = ((~pad).sum(1).float() / downsample).ceil().long()
    else:
        G, gmask = 1, torch.zeros(1, S)
        glen = torch.ones(B, dtype=torch.long)
    x = dec.transpose(0, 1).unsqueeze(0)                # [1, U, B, D]  (:836 + TransformerJointerLayer.forward :755-756)
    for li in range(layers):
        pre = f"{prefix}{li}."
        lin = lambda t, n: F.linear(t, P[pre + n + ".weight"], P[pre + n + ".bias"])                       # noqa: E731
        ln = lambda t, n: F.layer_norm(t, (D,), P[pre + n + ".weight"], P[pre + n + ".bias"], 1e-5)       # noqa: E731
        residual = x
        h = ln(x, "attn_layer_norm") if normalize_before else x
        # ExpandMultiheadAttention.forward (:642-715)
        Gin = h.shape[0]
        q = lin(h, "enc_attn.q_proj").view(Gin * U, B * heads, hd).transpose(0, 1) * hd ** -0.5
        k = lin(enc, "enc_attn.k_proj").view(S, B * heads, hd).transpose(0, 1)
        v = lin(enc, "enc_attn.v_proj").view(S, B * heads, hd).transpose(0, 1)
        w = torch.bmm(q, k.transpose(1, 2)).view(B, heads, Gin, U, S)
        w = w.masked_fill(pad.view(B, 1, 1, 1, S), float("-inf"))
        w = w + gmask.view(1, 1, G, 1, S)               # broadcasts Gin == 1 over the G groups (:700-705)
        p = torch.softmax(w.float(), dim=-1).view(B * heads, G, U, S)
        o = torch.einsum("bgts,bsd->bgtd", p, v).view(B, heads, G, U, hd).permute(2, 3, 0, 1, 4).reshape(G, U, B, D)
        o = lin(o, "enc_attn.out_proj")
        x = o + residual
        if not normalize_before:
            x = ln(x, "attn_layer_norm")
        residual = x
        h = ln(x, "final_layer_norm") if normalize_before else x
        h = lin(torch.relu(lin(h, "fc1")), "fc2")
        x = h + residual
        if not normalize_before:
            x = ln(x, "final_layer_norm")
    return x.permute(2, 0, 1, 3), glen                  # gxtxbxd -> bxgxtxd (:848-850)
