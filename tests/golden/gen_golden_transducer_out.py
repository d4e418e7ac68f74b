#!/usr/bin/env python3
"""Records data-only fixtures of the reference's CAAT loss head TransducerOut.train_step
(rain/layers/attention_transducer.py:289-408) at delay_scale = 0, run through oracle/ref_import.load_transducer_out(): the
reference's own class source and label_smoothed_nll_loss executed from where they lie, with DelayTLoss bound to the
reference's CPU transducer compiled into oracle/_ref (the CUDA-only delay term is the one thing left out).
One batch, evaluated with tokens_per_step giving 1 and 3 micro-batches, label smoothing 0.1, with and without a loss scaler:
the four losses, d x and d W.   tests/golden/transducer_out.npz   (container only; the fixtures travel, the reference does not)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import ref_import  # noqa: E402

H = ref_import.load_transducer_out()
torch.manual_seed(7)
B, T, U, d, V = 5, 9, 6, 32, 24                     # joint states [B, T, U, d]; targets [B, U-1]; U = max target length + 1
x0 = (torch.randn(B, T, U, d) * 0.7).to(torch.bfloat16).float()      # bf16-representable: a bf16 implementation sees the same numbers
W0 = (torch.randn(V, d) * 0.3).to(torch.bfloat16).float()
src_len = torch.tensor([9, 7, 9, 4, 6])
tgt_len = torch.tensor([5, 3, 4, 5, 2])
targets = torch.randint(2, V, (B, U - 1))
for b in range(B):
    targets[b, tgt_len[b]:] = 1                     # pad = 1 behind each target (attention_transducer.py:299, ignore_index)
out = {"x": x0.numpy(), "W": W0.numpy(), "targets": targets.numpy(), "src_len": src_len.numpy(), "tgt_len": tgt_len.numpy(),
       "cfg": np.array([B, T, U, d, V])}


class Scaler:                                        # what train_step asks of its optional scaler (:397-398)
    def __init__(self, s):
        self.s = s

    def scale(self, loss):
        return loss * self.s


for tag, tps, scale in (("mb1", 20000, None), ("mb3", 2 * T * U, None), ("mb3_scaled", 2 * T * U, 8.0)):
    proj = torch.nn.Linear(d, V, bias=False)
    with torch.no_grad():
        proj.weight.copy_(W0)
    head = H.TransducerOut(proj, delay_scale=0.0, tokens_per_step=tps, blank=0, label_smoothing=0.1, delay_func="zero", pad=1,
                           ce_scale=1.0, temperature=1.0)
    x = x0.clone().requires_grad_(True)
    res = head.train_step(x, targets, src_len, tgt_len, scaler=Scaler(scale) if scale else None)
    n_mb = len(x0.split(max(tps // (T * U), 1)))
    out.update({f"{tag}.loss": np.float64(res["loss"]), f"{tag}.loss_prob": np.float64(res["loss_prob"]),
                f"{tag}.nll_loss": np.float64(res["nll_loss"]), f"{tag}.sample_size": np.int64(res["sample_size"]),
                f"{tag}.dx": x.grad.numpy().copy(), f"{tag}.dW": proj.weight.grad.numpy().copy(),
                f"{tag}.tokens_per_step": np.int64(tps), f"{tag}.micro_batches": np.int64(n_mb),
                f"{tag}.loss_scale": np.float64(scale or 1.0)})
    print(tag, "micro-batches", n_mb, {k: float(v) for k, v in res.items()})
np.savez_compressed(os.path.join(HERE, "transducer_out.npz"), **out)
print("wrote transducer_out.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "transducer_out.npz")) // 1024, "KiB")
