#!/usr/bin/env python3
"""Build-container only: time ONE training step (fwd + loss + bwd, fp32, all host cores) of the reference itself
(imported through oracle/ref_import.py) and of the oracle restatement on cfgA (2 x 160 000 samples), same weights, and print
their ratio - the number BASELINE.md records so that the GPU box's `cpu_baseline` (kind "port") can be read as a
reference time.  Needs /root/reference."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import  # noqa: E402
import w2vs_oracle as O  # noqa: E402

torch.set_num_threads(os.cpu_count())
ref = ref_import.load()
cfg = ref_import.make_cfg(ref, context_type="constant", encoder_layerdrop=0.0, dropout_input=0.0, dropout_features=0.0)
cfg.dropout = cfg.attention_dropout = 0.0
torch.manual_seed(1); np.random.seed(1)
model = ref.Wav2VecSModel(cfg).train()
B, L = 2, 160000
src = torch.randn(B, L, generator=torch.Generator().manual_seed(1234))


def ref_step():
    model.zero_grad()
    np.random.seed(3); torch.manual_seed(3)
    out = model(src)
    logits = model.get_logits(out).float()
    loss = torch.nn.functional.cross_entropy(logits, torch.zeros(logits.shape[0], dtype=torch.long), reduction="sum")
    ex = model.get_extra_losses(out)
    loss = loss + 0.1 * ex[0] * logits.shape[0] + 10.0 * ex[1] * logits.shape[0]
    loss.backward()
    return float(loss)


ocfg = O.OracleCfg()
P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items() if v.dtype == torch.float32 and v.numel() > 1}
T = O.conv_out_lengths(L, ocfg.conv_layers)[-1]
np.random.seed(3)
mask = torch.from_numpy(O.compute_mask_indices((B, T), None, 0.65, 10, "static", 0, min_masks=2))
M = int(mask[0].sum())
torch.manual_seed(3)
neg = O.sample_negative_indices(B, M, 100)
noise = -torch.empty(B * M * 2, 320).exponential_().log()


def oracle_step():
    for p in P.values():
        p.grad = None
    out = O.forward_loss(P, src, ocfg, mask_indices=mask, neg_idx=neg, main_context=16, right_context=8, tau=2.0, gumbel_noise=noise)
    out["loss"].backward()
    return float(out["loss"])


def med(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


tr, to = med(ref_step), med(oracle_step)
print({"cores": os.cpu_count(), "reference_s_per_step": round(tr, 2), "oracle_s_per_step": round(to, 2),
       "oracle_over_reference": round(to / tr, 3), "reference_audio_s_per_s": round(B * L / 16000 / tr, 2)})
