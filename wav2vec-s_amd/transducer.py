"""Transducer losses on the HIP kernels.

Host-side mirror of the reference's PyTorch front end ``warp_transducer/pytorch_binding/warprnnt_pytorch``:
``rnnt.py`` (``RNNTLoss``, ``rnnt_loss``, argument checks :107-140) and ``delay_transducer.py`` (``DelayTLoss``,
``delay_transducer_loss``, the three delay-cost builders :96-134), the loss the streaming ST models train with
(``rain/layers/attention_transducer.py:42, 307-311, 389-391``: ``from warprnnt_pytorch import DelayTLoss``).

Same class / function names, arguments, return values and error behaviour.  What differs by design:
* the forward launches the denominators and both lattice recursions and leaves the three costs ON THE DEVICE (the
  reference copies them to the host and synchronises the stream inside every call, ``delay_transducer.h:366-368``);
* the gradient kernel runs in ``backward`` and multiplies by the incoming gradient in the same pass (the reference
  writes the gradient in forward and multiplies the whole [B,T,U,V] tensor again in Python, ``:86-90``);
* there is no CPU path: CPU tensors raise (the reference's ``DelayTLoss`` raises too, ``:51-52``).
"""
import ctypes as C

import torch
from torch.autograd import Function
from torch.nn import Module

from . import _lib
from ._lib import W2vsError


class RnntOptions(C.Structure):
    """struct rnntOptions (include/w2vs_rnnt.h; warp_transducer/include/rnnt.h:44-66), passed by value."""
    _fields_ = [("loc", C.c_int), ("num_threads", C.c_uint), ("stream", C.c_void_p), ("blank_label", C.c_int),
                ("maxT", C.c_int), ("maxU", C.c_int), ("batch_first", C.c_bool)]


_SIGS_DONE = False


def _rnnt_lib():
    global _SIGS_DONE
    lib = _lib.load()
    if not _SIGS_DONE:
        vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
        lib.rnntGetStatusString.restype = C.c_char_p
        lib.rnntGetStatusString.argtypes = [i32]
        lib.get_workspace_size.argtypes = [i32, i32, i32, C.c_bool, C.POINTER(C.c_size_t), C.c_size_t]
        lib.get_delay_workspace_size.argtypes = [i32, i32, i32, C.c_bool, C.POINTER(C.c_size_t), C.c_size_t]
        lib.compute_rnnt_loss.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, RnntOptions]
        lib.compute_rnnt_delay_loss.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, f32, f32, RnntOptions]
        lib.w2vs_rnnt_forward_async.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, f32, RnntOptions, vp, C.c_int64]
        lib.w2vs_rnnt_backward_async.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, f32, f32, i32, vp, i32, f32,
                                                 RnntOptions, vp, C.c_int64]
        lib.w2vs_rnnt_delay_values.argtypes = [i32, vp, vp, vp, i32, i32, i32, vp]
        lib.w2vs_ls_ce_rows.argtypes = [vp, vp, vp, vp, C.c_int64, i32, i32, f32, f32, i32, vp]
        for n in ("get_workspace_size", "get_delay_workspace_size", "compute_rnnt_loss", "compute_rnnt_delay_loss",
                  "w2vs_rnnt_forward_async", "w2vs_rnnt_backward_async", "w2vs_rnnt_delay_values", "get_warprnnt_version",
                  "w2vs_ls_ce_rows"):
            getattr(lib, n).restype = C.c_int
        _SIGS_DONE = True
    return lib


RNNT_EXPORTS = ["get_warprnnt_version", "rnntGetStatusString", "compute_rnnt_loss", "compute_rnnt_loss_fp64", "get_workspace_size",
                "compute_rnnt_delay_loss", "get_delay_workspace_size", "w2vs_rnnt_forward_async",
                "w2vs_rnnt_backward_async", "w2vs_rnnt_delay_values", "w2vs_ls_ce_rows"]


def _check(rc, what):
    if rc != 0:
        raise W2vsError("%s failed: %s" % (what, _rnnt_lib().rnntGetStatusString(rc).decode()))


def _options(acts, blank):
    return RnntOptions(1, 0, torch.cuda.current_stream(acts.device).cuda_stream, int(blank), acts.size(1), acts.size(2),
                       True)


def workspace_bytes(maxT, maxU, minibatch, delay=True):
    lib = _rnnt_lib()
    n = C.c_size_t(0)
    fn = lib.get_delay_workspace_size if delay else lib.get_workspace_size
    _check(fn(maxT, maxU, minibatch, True, C.byref(n), 4), "get_workspace_size")
    return n.value


# ---- argument checks, same messages as warprnnt_pytorch/rnnt.py:107-140 and delay_transducer.py:11-41
def check_type(var, t, name):
    if var.dtype is not t:
        raise TypeError("{} must be {}".format(name, t))


def check_contiguous(var, name):
    if not var.is_contiguous():
        raise ValueError("{} must be contiguous".format(name))


def check_dim(var, dim, name):
    if len(var.shape) != dim:
        raise ValueError("{} must be {}D".format(name, dim))


def certify_inputs(log_probs, labels, lengths, label_lengths, delay_values=None, strict_lengths=True):
    check_type(labels, torch.int32, "labels")
    check_type(label_lengths, torch.int32, "label_lengths")
    check_type(lengths, torch.int32, "lengths")
    check_contiguous(log_probs, "log_probs")
    check_contiguous(labels, "labels")
    check_contiguous(label_lengths, "label_lengths")
    check_contiguous(lengths, "lengths")
    if delay_values is not None:
        check_type(delay_values, torch.float32, "delay_values")
        check_contiguous(delay_values, "delay_values")
    if lengths.shape[0] != log_probs.shape[0]:
        raise ValueError("must have a length per example.")
    if label_lengths.shape[0] != log_probs.shape[0]:
        raise ValueError("must have a label length per example.")
    check_dim(log_probs, 4, "log_probs")
    check_dim(labels, 2, "labels")
    check_dim(lengths, 1, "lenghts")
    check_dim(label_lengths, 1, "label_lenghts")
    if delay_values is not None:
        check_dim(delay_values, 3, "delay_values")
    if strict_lengths:                                   # rnnt.py:134-140 (the delay front end dropped this check)
        T, U = log_probs.shape[1:3]
        if T != int(torch.max(lengths)):
            raise ValueError("Input length mismatch")
        if U != int(torch.max(label_lengths)) + 1:
            raise ValueError("Output length mismatch")
    # what the kernels additionally rely on (the reference would read out of bounds instead)
    if log_probs.dtype is not torch.float32:
        raise TypeError("acts must be torch.float32")
    if labels.shape[1] != log_probs.shape[2] - 1:
        raise ValueError("labels must be [B, U-1] for acts [B, T, U, V]")
    if delay_values is not None and tuple(delay_values.shape) != tuple(log_probs.shape[:3]):
        raise ValueError("delay_values must be [B, T, U]")


class _Transducer(Function):
    """One autograd node for both losses (delay_values None = plain RNN-T)."""

    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, delay_values, delay_scale, blank, temperature, reduction,
                consistent_delay_index):
        if not acts.is_cuda:
            raise NotImplementedError("only gpu version now")            # delay_transducer.py:51-52
        lib = _rnnt_lib()
        dev = acts.device
        labels, act_lens, label_lens = labels.to(dev), act_lens.to(dev), label_lens.to(dev)
        B, T, U, V = acts.shape
        x = acts.detach()
        ws = torch.empty(workspace_bytes(T, U, B, delay_values is not None) // 4, dtype=torch.float32, device=dev)
        costs = torch.empty(3, B, dtype=torch.float32, device=dev)
        opt = _options(x, blank)
        dvp = delay_values.data_ptr() if delay_values is not None else None
        _check(lib.w2vs_rnnt_forward_async(x.data_ptr(), labels.data_ptr(), label_lens.data_ptr(), act_lens.data_ptr(), dvp,
                                           V, B, costs.data_ptr(), ws.data_ptr(), float(delay_scale), opt, None, 0),
               "w2vs_rnnt_forward_async")
        ctx.saved = (x, labels, act_lens, label_lens, delay_values, ws)
        ctx.args = (float(delay_scale), int(blank), float(temperature), reduction, bool(consistent_delay_index))
        loss_rnnt, loss_delay, loss_total = costs[0], costs[1], costs[2]
        if reduction in ["sum", "mean"]:
            loss_rnnt, loss_delay, loss_total = loss_rnnt.sum(), loss_delay.sum(), loss_total.sum()
            if reduction == "mean":
                loss_rnnt, loss_delay, loss_total = loss_rnnt / B, loss_delay / B, loss_total / B
        return loss_total, loss_rnnt, loss_delay

    @staticmethod
    def backward(ctx, grad_output, g2, g3):
        x, labels, act_lens, label_lens, delay_values, ws = ctx.saved
        delay_scale, blank, temperature, reduction, consistent = ctx.args
        lib = _rnnt_lib()
        B, T, U, V = x.shape
        grads = torch.empty_like(x)
        up = grad_output.detach().to(torch.float32).reshape(-1).contiguous()
        if up.numel() not in (1, B):
            raise W2vsError("transducer backward: the incoming gradient must be a scalar or one value per sample")
        opt = _options(x, blank)
        dvp = delay_values.data_ptr() if delay_values is not None else None
        _check(lib.w2vs_rnnt_backward_async(x.data_ptr(), grads.data_ptr(), labels.data_ptr(), label_lens.data_ptr(),
                                            act_lens.data_ptr(), dvp, V, B, ws.data_ptr(), delay_scale, temperature,
                                            1 if consistent else 0, up.data_ptr(), up.numel(),
                                            1.0 / B if reduction == "mean" else 1.0, opt, None, 0), "w2vs_rnnt_backward_async")
        return grads, None, None, None, None, None, None, None, None, None


def _delay_values(kind, acts, src_lens, tgt_lens):
    if not acts.is_cuda:
        raise NotImplementedError("only gpu version now")
    B, S, T = acts.shape[:3]
    out = torch.empty(B, S, T, dtype=torch.float32, device=acts.device)
    src = src_lens.to(device=acts.device, dtype=torch.int32).contiguous()
    tgt = tgt_lens.to(device=acts.device, dtype=torch.int32).contiguous()
    _check(_rnnt_lib().w2vs_rnnt_delay_values(kind, src.data_ptr(), tgt.data_ptr(), out.data_ptr(), B, S, T,
                                              torch.cuda.current_stream(acts.device).cuda_stream), "w2vs_rnnt_delay_values")
    return out


def delay_cost_zero(acts, src_lens, tgt_lens):
    """delay_transducer.py:96-101: reading frame s costs s / src_len, whatever the label."""
    return _delay_values(0, acts, src_lens, tgt_lens)


def delay_cost_diagonal(acts, src_lens, tgt_lens):
    """delay_transducer.py:118-134: distance from the diagonal."""
    return _delay_values(1, acts, src_lens, tgt_lens)


def delay_cost_diag_positive(acts, src_lens, tgt_lens):
    """delay_transducer.py:103-116: only lagging behind the diagonal costs."""
    return _delay_values(2, acts, src_lens, tgt_lens)


def delay_transducer_loss(acts, labels, act_lens, label_lens, delay_values, delay_scale=1.0, temperature=1.0, blank=0,
                          reduction="sum", consistent_delay_index=False):
    """delay_transducer.py:89-94.  Returns (loss_total, loss_rnnt, loss_delay)."""
    certify_inputs(acts, labels, act_lens, label_lens, delay_values, strict_lengths=False)
    return _Transducer.apply(acts, labels, act_lens, label_lens, delay_values, delay_scale, blank, temperature, reduction,
                             consistent_delay_index)


class DelayTLoss(Module):
    """delay_transducer.py:137-177: ``blank``, ``delay_scale``, ``temperature``, ``reduction`` ('none' | 'mean' | 'sum'),
    ``delay_func`` ('zero' | 'diagonal' | 'diag_positive').  ``consistent_delay_index`` (an addition, default False =
    the reference's behaviour): see include/w2vs_rnnt.h flags bit 0."""

    def __init__(self, blank=0, delay_scale=1.0, temperature=1.0, reduction="sum", delay_func="zero",
                 consistent_delay_index=False):
        super().__init__()
        self.delay_scale, self.blank, self.reduction, self.temperature = delay_scale, blank, reduction, temperature
        delay_funcs = {"zero": delay_cost_zero, "diagonal": delay_cost_diagonal, "diag_positive": delay_cost_diag_positive}
        if delay_func not in delay_funcs:
            raise NotImplementedError(f"{delay_func} not implemented")
        self.delay_func = delay_funcs[delay_func]
        self.consistent_delay_index = consistent_delay_index

    def forward(self, acts, labels, act_lens, label_lens):
        with torch.no_grad():
            delay_values = self.delay_func(acts, act_lens, label_lens)
        return delay_transducer_loss(acts, labels, act_lens, label_lens, delay_values, self.delay_scale, self.temperature,
                                     self.blank, self.reduction, self.consistent_delay_index)


def rnnt_loss(acts, labels, act_lens, label_lens, blank=0, reduction="mean"):
    """rnnt.py:54-72.  Raw activations in (the log-softmax is fused, as in the reference's GPU path)."""
    if not acts.is_cuda:
        raise W2vsError("rnnt_loss runs on an MI355X only (there is no CPU path)")
    certify_inputs(acts, labels, act_lens, label_lens)
    total, _, _ = _Transducer.apply(acts, labels, act_lens, label_lens, None, 0.0, blank, 1.0, reduction, False)
    return total.unsqueeze(-1) if reduction in ["sum", "mean"] else total      # rnnt.py:36: shape [1]


class RNNTLoss(Module):
    """rnnt.py:75-104."""

    def __init__(self, blank=0, reduction="mean"):
        super().__init__()
        self.blank, self.reduction = blank, reduction

    def forward(self, acts, labels, act_lens, label_lens):
        return rnnt_loss(acts, labels, act_lens, label_lens, self.blank, self.reduction)


# ---------------------------------------------------------------------------------------------------------------------
# The loss head of the CAAT transducer: output projection + delay-transducer loss + label-smoothed CE on the last frame
# ---------------------------------------------------------------------------------------------------------------------
class TransducerOut(Module):
    """rain/layers/attention_transducer.py:289-456.  "This module is special": ``train_step`` runs the forward AND the
    backward of the head itself, in micro-batches bounded by ``tokens_per_step`` lattice cells, accumulates the
    gradient of ``output_proj`` and pushes the gradient of the joint states ``x`` into the rest of the network with ONE
    ``autograd.backward(x, input_grads)`` (:401).

    Same constructor arguments, same methods and result dictionaries.  The computation is explicit HIP launches instead
    of an autograd graph per micro-batch: logits = x W^T (bf16 MFMA GEMM, fp32 out) -> transducer forward -> gradient
    rows written directly as bf16 with the loss scale folded in -> d x = d logits W and d W += d logits^T x (bf16 GEMMs,
    fp32 accumulation); the cross-entropy branch likewise (``w2vs_ls_ce_rows``).  Losses stay on the device.
    ``output_proj`` must be a bias-free ``nn.Linear`` (what the reference builds, :851) with widths multiple of 8."""

    def __init__(self, output_proj, delay_scale=1.0, tokens_per_step=20000, blank=0, smoothing=0.0, label_smoothing=0.1,
                 delay_func="zero", pad=1, ce_scale=1.0, temperature=1.0):
        super().__init__()
        self.rnnt_loss = DelayTLoss(blank=blank, delay_scale=delay_scale, temperature=temperature, reduction="sum",
                                    delay_func=delay_func)
        self.output_proj = output_proj
        self.vocab_size = output_proj.weight.shape[0]
        self.delay_scale, self.tokens_per_step, self.smoothing, self.pad = delay_scale, tokens_per_step, smoothing, pad
        self.label_smoothing, self.ce_scale = label_smoothing, ce_scale
        self.blank, self.temperature = blank, temperature
        if getattr(output_proj, "bias", None) is not None:
            raise W2vsError("TransducerOut: output_proj must be bias-free (rain/layers/attention_transducer.py:851)")

    # ---- plumbing
    def _w16(self):
        w = self.output_proj.weight.detach()
        return (w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)).contiguous()

    @staticmethod
    def _logits(x2, w16):
        from . import ops
        R, d = x2.shape
        V = w16.shape[0]
        out = torch.empty(R, V, dtype=torch.float32, device=x2.device)
        ops.gemm_nt(x2, w16, M=R, N=V, K=d, lda=d, ldb=d, ldc=V, out_f32=out, epi=_lib.EPI_F32)
        return out

    def forward(self, x):
        """:320-321 ``self.output_proj(x)`` (model dtype in and out)."""
        if not x.is_cuda:
            raise W2vsError("TransducerOut runs on an MI355X only (there is no CPU path)")
        shp = x.shape
        x2 = x.detach().to(torch.bfloat16).reshape(-1, shp[-1]).contiguous()
        return self._logits(x2, self._w16()).view(*shp[:-1], self.vocab_size).to(x.dtype)

    def _chunk(self, x, targets, slen, tlen, w16, wt16, dw32, loss_scale, slen_h, tlen_h):
        """One micro-batch: returns (loss_total, loss_prob, loss_delay, nll) as device scalars and d x (bf16), or None
        for d x when ``dw32`` is None (evaluation).  Only the lattice cells inside each sample's T_b x (U_b + 1) go
        through the projection, the loss kernels and the two gradient GEMMs (``cell_index`` of the async API): the padding
        of a ragged batch costs nothing."""
        from . import ops
        import numpy as np
        lib = _rnnt_lib()
        B, T, U, d = x.shape
        V = self.vocab_size
        dev = x.device
        train = dw32 is not None
        x2 = x.reshape(-1, d)
        tt = np.arange(T)[None, :, None] < np.clip(slen_h, 1, T)[:, None, None]
        uu = np.arange(U)[None, None, :] < np.clip(tlen_h + 1, 1, U)[:, None, None]
        cells_h = torch.from_numpy(np.flatnonzero(tt & uu).astype(np.int32)).pin_memory()
        n = int(cells_h.numel())
        cells = cells_h.to(dev, non_blocking=True)
        xc = ops.gather_rows(x2, cells, n)                                    # [n, d] bf16
        logits = self._logits(xc, w16)                                        # [n, V] fp32 (:375-376)
        lab, sl, tl = targets.int().contiguous(), slen.int().contiguous(), tlen.int().contiguous()
        dv = self.rnnt_loss.delay_func(x, sl, tl)                             # [B, T, U] (only the shape of x is used)
        ws = torch.empty(workspace_bytes(T, U, B, True) // 4, dtype=torch.float32, device=dev)
        costs = torch.empty(3, B, dtype=torch.float32, device=dev)
        opt = RnntOptions(1, 0, torch.cuda.current_stream(dev).cuda_stream, int(self.blank), T, U, True)
        _check(lib.w2vs_rnnt_forward_async(logits.data_ptr(), lab.data_ptr(), tl.data_ptr(), sl.data_ptr(), dv.data_ptr(), V, B,
                                           costs.data_ptr(), ws.data_ptr(), float(self.delay_scale), opt, cells.data_ptr(), n),
               "w2vs_rnnt_forward_async")
        dx = None
        if train:
            dl = torch.empty(n, V, dtype=torch.bfloat16, device=dev)
            _check(lib.w2vs_rnnt_backward_async(logits.data_ptr(), dl.data_ptr(), lab.data_ptr(), tl.data_ptr(), sl.data_ptr(),
                                                dv.data_ptr(), V, B, ws.data_ptr(), float(self.delay_scale),
                                                float(self.temperature), 2, None, 0, float(loss_scale), opt, cells.data_ptr(), n),
                   "w2vs_rnnt_backward_async")
            dxc = ops.linear_dgrad(dl, wt16)                                  # [n, d] bf16
            ops.linear_wgrad(dl, xc, dw32)
            dx = torch.zeros(B * T * U, d, dtype=torch.bfloat16, device=dev)
            ops.gather_rows(dxc, cells, n, scatter=True, out=dx)
            del dl, dxc
        del logits, xc
        # ---- cross-entropy on the hidden state at each sample's last frame (:339-360)
        bidx = torch.arange(B, device=dev)
        last_h = x[bidx, (slen.long() - 1).clamp_(0, T - 1)][:, :-1].contiguous()      # [B, U-1, d]
        R2 = B * (U - 1)
        logits2 = self._logits(last_h.view(R2, d), w16)
        sums = torch.zeros(2, dtype=torch.float32, device=dev)
        tgt = targets.int().contiguous()
        if tuple(tgt.shape) != (B, U - 1):
            raise W2vsError("TransducerOut: targets must be [B, U-1] for joint states [B, T, U, d]")
        dl2 = torch.empty(R2, V, dtype=torch.bfloat16, device=dev) if train else None
        _check(lib.w2vs_ls_ce_rows(logits2.data_ptr(), tgt.data_ptr(), dl2.data_ptr() if train else None, sums.data_ptr(), R2, V,
                                   int(self.pad), float(self.label_smoothing), float(self.ce_scale * loss_scale), 1,
                                   torch.cuda.current_stream(dev).cuda_stream), "w2vs_ls_ce_rows")
        if train:
            dh = ops.linear_dgrad(dl2, wt16).view(B, U - 1, d)
            ops.linear_wgrad(dl2, last_h.view(R2, d), dw32)
            dx4 = dx.view(B, T, U, d)
            dx4[bidx, (slen.long() - 1).clamp_(0, T - 1), :U - 1] += dh      # one (b, t) row per sample: no collisions
        loss = costs[2].sum() + self.ce_scale * sums[0]
        return (loss, costs[0].sum(), costs[1].sum(), sums[1]), dx

    def _run(self, x, targets, src_lengths, tgt_lengths, scaler, train):
        from . import ops
        import numpy as np
        if not x.is_cuda:
            raise W2vsError("TransducerOut runs on an MI355X only (there is no CPU path)")
        B, T, U, d = x.shape
        bsz_per_step = max(self.tokens_per_step // (T * U), 1)                 # :370, :421
        w16 = self._w16()
        wt16 = ops.transpose2d(w16) if train else None
        dw32 = torch.zeros(w16.shape, dtype=torch.float32, device=x.device) if train else None
        loss_scale = 1.0
        if scaler is not None and train:
            # the reference only calls scaler.scale(loss) (attention_transducer.py:397-398): a torch GradScaler has get_scale();
            # anything else is asked what it does to a one
            loss_scale = float(scaler.get_scale()) if hasattr(scaler, "get_scale") else float(scaler.scale(torch.ones(())))
        xs = x.detach()
        xs = (xs if xs.dtype == torch.bfloat16 else xs.to(torch.bfloat16)).contiguous()
        # the one host round trip of the step (the reference has one too: `.item()` on the token count, :402): the lengths decide
        # which lattice cells exist, and only those are projected
        slen_h, tlen_h = src_lengths.cpu().numpy().astype(np.int64), tgt_lengths.cpu().numpy().astype(np.int64)
        losses = [0, 0, 0, 0]
        grads = []
        for i in range(0, B, bsz_per_step):
            j = min(i + bsz_per_step, B)
            ls, dx = self._chunk(xs[i:j], targets[i:j], src_lengths[i:j], tgt_lengths[i:j], w16, wt16, dw32, loss_scale,
                                 slen_h[i:j], tlen_h[i:j])
            losses = [a + b.detach() for a, b in zip(losses, ls)]
            if train:
                grads.append(dx.view(j - i, T, U, d))
        if train:
            w = self.output_proj.weight
            g = dw32.to(w.dtype)
            w.grad = g if w.grad is None else w.grad + g
            torch.autograd.backward(x, torch.cat(grads, dim=0).to(x.dtype))    # :401
        ntokens = targets.ne(self.pad).sum().item()
        return {"loss": losses[0], "loss_prob": losses[1], "loss_delay": losses[2], "nll_loss": losses[3],
                "sample_size": ntokens}

    def train_step(self, x, targets, src_lengths, tgt_lengths, scaler=None):
        """:362-408.  x [B, T, U+1, d] joint states (requires grad), targets [B, U]."""
        return self._run(x, targets, src_lengths, tgt_lengths, scaler, True)

    def eval_step(self, x, targets, src_lengths, tgt_lengths):
        """:410-446."""
        with torch.no_grad():
            return self._run(x, targets, src_lengths, tgt_lengths, None, False)
