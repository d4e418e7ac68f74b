#!/usr/bin/env python3
"""Records data-only fixtures of the reference's CAAT joint network (rain/layers/attention_transducer.py:591-852), run
through oracle/ref_import.load_joiner() (the reference's own class source, executed from where it lies):
parameters, inputs, output, group lengths and every gradient of a fixed linear functional, for a pre-LN and a post-LN
net.  tests/golden/joiner.npz  (container only; the fixtures travel, the reference does not)."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import ref_import  # noqa: E402

J = ref_import.load_joiner()
out = {}
for tag, pre_ln, layers, ds in (("pre", True, 2, 8), ("post", False, 2, 5), ("offline", True, 1, -1)):
    torch.manual_seed({"pre": 1, "post": 2, "offline": 3}[tag])
    D, H = (128, 2) if tag == "pre" else (64, 1)       # head_dim 64, as in rain (jointer_embed_dim 256 / 4 heads)
    S, U, B = 37, 7, 2
    args = argparse.Namespace(jointer_embed_dim=D, jointer_attention_heads=H, transducer_downsample=ds, jointer_layers=layers,
                              attention_dropout=0.1, dropout=0.1, activation_dropout=0.1, activation_fn="relu",
                              encoder_normalize_before=pre_ln, jointer_ffn_embed_dim=2 * D, step_mode="constant")
    net = J.MHAJointNet(args).eval()
    with torch.no_grad():                         # LayerNorm affine and biases away from their trivial init
        for n, p in net.named_parameters():
            if "layer_norm" in n or n.endswith("bias"):
                p.add_(torch.randn_like(p) * 0.1)
            p.copy_(p.to(torch.bfloat16).float())  # bf16-representable parameters: a bf16 implementation sees the same numbers
    enc = torch.randn(S, B, D).to(torch.bfloat16).float().requires_grad_(True)
    dec = torch.randn(B, U, D).to(torch.bfloat16).float().requires_grad_(True)
    pad = torch.zeros(B, S, dtype=torch.bool)
    pad[1, 29:] = True
    x, glen = net({"encoder_out": [enc], "encoder_padding_mask": [pad]}, dec)
    w = torch.randn(x.shape)
    (x * w).sum().backward()
    out.update({f"{tag}.cfg": np.array([D, H, S, U, B, layers, ds, int(pre_ln)]), f"{tag}.enc": enc.detach().numpy(),
                f"{tag}.dec": dec.detach().numpy(), f"{tag}.pad": pad.numpy(), f"{tag}.x": x.detach().numpy(),
                f"{tag}.glen": glen.numpy(), f"{tag}.w": w.numpy(), f"{tag}.d_enc": enc.grad.numpy(),
                f"{tag}.d_dec": dec.grad.numpy()})
    for n, p in net.named_parameters():
        out[f"{tag}.P.{n}"] = p.detach().numpy()
        out[f"{tag}.G.{n}"] = p.grad.numpy()
np.savez_compressed(os.path.join(HERE, "joiner.npz"), **out)
print("wrote joiner.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "joiner.npz")) // 1024, "KiB")
