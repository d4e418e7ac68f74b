#!/usr/bin/env python3
"""Golden vectors for SURVEY.md section 8 row f1 (streaming / fine-tune encoder twin), recorded
from the REAL reference (``rain/layers/unidirect_w2v2_encoder.py``, imported through
``oracle/ref_import.load_rain``) in the build container.

    python tests/golden/gen_golden_stream.py

Outputs (data only):
    tests/golden/stream_twin.npz    BlockWiseWav2Vec2Model: post-LN, 2 small layers (layer_norm_num = 7), padded batch;
                                    full / is_infer outputs, padding masks, all parameter gradients
                                    of a fixed linear functional of the valid frames
    tests/golden/stream_online.npz  OnlineW2V2TransformerEncoder built from a checkpoint file:
                                    pre-LN large-style twin + encoder_proj; frozen and unfrozen
                                    (freeze_finetune_updates) gradients; reorder_encoder_out
"""
import argparse
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import  # noqa: E402

rain = ref_import.load_rain()

TINY = dict(
    # head_dim 64 and 8-aligned widths: the same fixture drives the HIP kernels (tests/test_stream_gpu.py)
    extractor_mode="layer_norm", encoder_embed_dim=128, encoder_ffn_embed_dim=256, encoder_attention_heads=2,
    final_dim=16, latent_vars=8, latent_groups=2, num_negatives=10, quantize_targets=True,
    conv_feature_layers="[(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64,2,2)] * 2",
    dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0, dropout_features=0.0,
    encoder_layerdrop=0.0, pos_type="sin", load_pretrained_model_from=None,
)


def seed_all(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def perturb(model, seed):
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias") or "layer_norm" in n or ".2.1." in n:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    return g


def padded_batch(g, B, L, lens):
    source = torch.randn(B, L, generator=g)
    pm = torch.zeros(B, L, dtype=torch.bool)
    for b, n in enumerate(lens):
        pm[b, n:] = True
        source[b, n:] = 0
    return source, pm


def functional(x, pad, g):
    """A fixed linear functional of the valid frames: sum(x * w * valid)."""
    w = torch.randn(x.shape, generator=g)
    valid = (~pad).transpose(0, 1).unsqueeze(-1).to(x.dtype)        # T x B x 1
    return (x * w * valid).sum(), w


def grads_of(model, out):
    for n, p in model.named_parameters():
        out["grad." + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        out["hasgrad." + n] = np.array([p.grad is not None])


def gen_twin():
    seed_all(11)
    over = dict(TINY, encoder_layers=2, main_context=8, right_context=4, feature_grad_mult=0.1)
    model = rain.BlockWiseWav2Vec2Model.build_model(argparse.Namespace(**over))
    g = perturb(model, 11)
    B, L = 2, 16000
    source, pm = padded_batch(g, B, L, [16000, 11000])
    out = {"source": source.numpy(), "padding_mask": pm.numpy()}
    model.train()                                           # dropouts are 0: train == eval numerically
    res = model(source, pm)
    x, pad = res["encoder_out"][0], res["encoder_padding_mask"][0]
    loss, w = functional(x, pad, g)
    model.zero_grad()
    loss.backward()
    out.update(x_full=x.detach().numpy(), pad_full=pad.numpy(), w=w.numpy(), loss=np.array([loss.item()]))
    grads_of(model, out)
    model.eval()
    with torch.no_grad():
        r1 = model(source, pm, None, False, True)            # streaming, unfinished: right context withheld
        r2 = model(source, pm, None, True, True)             # streaming, finished
        r3 = model(source[:, :9000])                         # no padding mask, a shorter prefix (odd T)
        r4 = model(source[:, :9000], None, None, False, True)
    out.update(x_infer=r1["encoder_out"][0].numpy(), pad_infer=r1["encoder_padding_mask"][0].numpy(),
               x_finished=r2["encoder_out"][0].numpy(), pad_finished=r2["encoder_padding_mask"][0].numpy(),
               x_prefix=r3["encoder_out"][0].numpy(), pad_prefix=r3["encoder_padding_mask"][0].numpy(),
               x_prefix_infer=r4["encoder_out"][0].numpy(), pad_prefix_infer=r4["encoder_padding_mask"][0].numpy())
    for n, p in model.state_dict().items():
        out["param." + n] = p.numpy()
    out["cfg_json"] = np.frombuffer(repr(over).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "stream_twin.npz"), **out)
    print("stream_twin", tuple(x.shape), "loss", loss.item(), "arrays", len(out))


def gen_online():
    seed_all(12)
    over = dict(TINY, encoder_layers=3, layer_norm_first=True, conv_bias=True, feature_grad_mult=1.0,
                main_context=16, right_context=8)           # the checkpoint's own contexts: overridden below
    pre = rain.BlockWiseWav2Vec2Model.build_model(argparse.Namespace(**over))
    g = perturb(pre, 12)
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "ckpt.pt")
    torch.save({"args": None, "cfg": {"model": dict(over)}, "model": pre.state_dict()}, path)
    args = argparse.Namespace(w2v2_model_path=path, main_context=4, right_context=2, use_linear_layer=True,
                              encoder_embed_dim=48, freeze_finetune_updates=5)
    enc = rain.OnlineW2V2TransformerEncoder(args)
    with torch.no_grad():
        enc.encoder_proj.weight.add_(torch.randn(enc.encoder_proj.weight.shape, generator=g) * 0.1)
        enc.encoder_proj.bias.add_(torch.randn(enc.encoder_proj.bias.shape, generator=g) * 0.05)
    B, L = 3, 12400
    lens = torch.tensor([12400, 9000, 6001])
    source, _ = padded_batch(g, B, L, lens.tolist())
    out = {"source": source.numpy(), "src_lengths": lens.numpy(), "init_frames": np.array([enc.init_frames]),
           "step_frames": np.array([enc.step_frames])}
    enc.train()
    for tag, upd in (("frozen", 0), ("tuned", 5)):
        enc.set_num_updates(upd)
        enc.zero_grad()
        res = enc(source, lens)
        x, pad = res["encoder_out"][0], res["encoder_padding_mask"][0]
        gg = torch.Generator().manual_seed(77)
        loss, w = functional(x, pad, gg)
        loss.backward()
        out.update({f"{tag}.x": x.detach().numpy(), f"{tag}.pad": pad.numpy(), f"{tag}.loss": np.array([loss.item()]),
                    "w": w.numpy()})
        for n, p in enc.named_parameters():
            out[f"{tag}.grad.{n}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
            out[f"{tag}.hasgrad.{n}"] = np.array([p.grad is not None])
    enc.eval()
    with torch.no_grad():
        r = enc(source, lens, None, False, True)
        order = torch.tensor([2, 0, 0, 1])
        ro = enc.reorder_encoder_out(r, order)
    out.update({"infer.x": r["encoder_out"][0].numpy(), "infer.pad": r["encoder_padding_mask"][0].numpy(),
                "reorder.order": order.numpy(), "reorder.x": ro["encoder_out"][0].numpy(),
                "reorder.pad": ro["encoder_padding_mask"][0].numpy()})
    for n, p in enc.state_dict().items():
        out["param." + n] = p.numpy()
    out["cfg_json"] = np.frombuffer(repr(over).encode(), dtype=np.uint8)
    out["args_json"] = np.frombuffer(repr(dict(main_context=4, right_context=2, use_linear_layer=True,
                                                 encoder_embed_dim=48, freeze_finetune_updates=5)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "stream_online.npz"), **out)
    print("stream_online", tuple(x.shape), "arrays", len(out))


if __name__ == "__main__":
    gen_twin()
    gen_online()
