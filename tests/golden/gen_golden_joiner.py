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
# ---- the decoding path: incremental_state (attention_transducer.py:607-674) with downsample = -1, as TransducerMHADecoder.forward
# sets it (:901-902).  Four calls on one state dictionary: fresh; SAME prefix length with DIFFERENT frames (the cached
# projections of the first call are reused - the reference only compares lengths); a longer prefix (recomputed); after a
# beam reorder that duplicates hypothesis 1.
torch.manual_seed(9)
D, H, layers = 128, 2, 2
args = argparse.Namespace(jointer_embed_dim=D, jointer_attention_heads=H, transducer_downsample=-1, jointer_layers=layers,
                          attention_dropout=0.1, dropout=0.1, activation_dropout=0.1, activation_fn="relu",
                          encoder_normalize_before=True, jointer_ffn_embed_dim=2 * D, step_mode="constant")
net = J.MHAJointNet(args).eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if "layer_norm" in n or n.endswith("bias"):
            p.add_(torch.randn_like(p) * 0.1)
        p.copy_(p.to(torch.bfloat16).float())
    rb = lambda *shape: torch.randn(*shape).to(torch.bfloat16).float()       # noqa: E731
    B = 2
    encA, encB, encC = rb(20, B, D), rb(20, B, D), rb(28, B, D)
    dec1, dec2, dec3 = rb(B, 3, D), rb(B, 4, D), rb(B, 5, D)
    padA, padC = torch.zeros(B, 20, dtype=torch.bool), torch.zeros(B, 28, dtype=torch.bool)
    state = {}
    o1, _ = net({"encoder_out": [encA], "encoder_padding_mask": [padA]}, dec1, incremental_state=state)
    o2, _ = net({"encoder_out": [encB], "encoder_padding_mask": [padA]}, dec2, incremental_state=state)
    o3, _ = net({"encoder_out": [encC], "encoder_padding_mask": [padC]}, dec3, incremental_state=state)
    order = torch.tensor([1, 1, 0])          # a different size than the cached batch: the reference reorders only then (:617-619)
    for layer in net.layers:
        layer.enc_attn.reorder_incremental_state(state, order)
    o4, _ = net({"encoder_out": [encC.index_select(1, order)], "encoder_padding_mask": [padC.index_select(0, order)]}, dec3.index_select(0, order),
                incremental_state=state)
out.update({"inc.cfg": np.array([D, H, layers, B]), "inc.encA": encA.numpy(), "inc.encB": encB.numpy(), "inc.encC": encC.numpy(),
            "inc.dec1": dec1.numpy(), "inc.dec2": dec2.numpy(), "inc.dec3": dec3.numpy(), "inc.o1": o1.numpy(), "inc.o2": o2.numpy(),
            "inc.o3": o3.numpy(), "inc.o4": o4.numpy()})
for n, p in net.named_parameters():
    out[f"inc.P.{n}"] = p.detach().numpy()
np.savez_compressed(os.path.join(HERE, "joiner.npz"), **out)
print("wrote joiner.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "joiner.npz")) // 1024, "KiB")
