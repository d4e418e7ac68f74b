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
        lib.w2vs_rnnt_forward_async.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, f32, RnntOptions]
        lib.w2vs_rnnt_backward_async.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, f32, f32, i32, vp, i32, f32,
                                                 RnntOptions]
        lib.w2vs_rnnt_delay_values.argtypes = [i32, vp, vp, vp, i32, i32, i32, vp]
        for n in ("get_workspace_size", "get_delay_workspace_size", "compute_rnnt_loss", "compute_rnnt_delay_loss",
                  "w2vs_rnnt_forward_async", "w2vs_rnnt_backward_async", "w2vs_rnnt_delay_values", "get_warprnnt_version"):
            getattr(lib, n).restype = C.c_int
        _SIGS_DONE = True
    return lib


RNNT_EXPORTS = ["get_warprnnt_version", "rnntGetStatusString", "compute_rnnt_loss", "get_workspace_size",
                "compute_rnnt_delay_loss", "get_delay_workspace_size", "w2vs_rnnt_forward_async",
                "w2vs_rnnt_backward_async", "w2vs_rnnt_delay_values"]


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
                                           V, B, costs.data_ptr(), ws.data_ptr(), float(delay_scale), opt),
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
                                            1.0 / B if reduction == "mean" else 1.0, opt), "w2vs_rnnt_backward_async")
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
