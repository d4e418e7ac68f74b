#!/usr/bin/env python3
"""BASELINE config 5 shape on one MI355X: streaming-ST fine-tune step = wav2vec-S base encoder twin (row f1) ->
encoder_proj 768 -> 256 -> CAAT joiner (6 layers, row f4) over decoder states -> TransducerOut (projection, delay
transducer loss, cross entropy) -> backward through all of it.  Synthetic MuST-C-shaped batch: 8 utterances of 6 s
(96 000 samples -> 299 frames), 47 target tokens, downsample 16 (19 groups), vocabulary 8000.  The autoregressive
text decoder that produces the decoder states (a fairseq TransformerDecoder, outside SURVEY section 8) is replaced by
a random-state input; its cost is not measured.

    python tools/bench_caat.py
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--samples", type=int, default=96000)
    ap.add_argument("--U", type=int, default=48)
    ap.add_argument("--V", type=int, default=8000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "caat_bench.json"))
    a = ap.parse_args(argv)
    from wav2vec_s_amd import joiner, streaming, transducer
    BF = torch.bfloat16
    torch.manual_seed(0)
    kw = dict(extractor_mode="layer_norm", encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072,
              encoder_attention_heads=12, final_dim=256, quantize_targets=True, feature_grad_mult=0.1, dropout=0.1,
              attention_dropout=0.1, dropout_input=0.0, dropout_features=0.0, encoder_layerdrop=0.0,
              conv_feature_layers="[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] * 2", main_context=16, right_context=8,
              pos_type="sin", load_pretrained_model_from=None)
    enc = streaming.BlockWiseWav2Vec2Model.build_model(argparse.Namespace(**kw)).to(BF).cuda().train()
    proj = streaming.HipLinear(768, 256).to(BF).cuda()
    jargs = argparse.Namespace(jointer_embed_dim=256, jointer_attention_heads=4, transducer_downsample=16, jointer_layers=6,
                               attention_dropout=0.1, dropout=0.1, activation_dropout=0.1, activation_fn="relu",
                               encoder_normalize_before=True, jointer_ffn_embed_dim=1024, step_mode="constant")
    jn = joiner.MHAJointNet(jargs).to(BF).cuda().train()
    head = transducer.TransducerOut(torch.nn.Linear(256, a.V, bias=False).to(BF).cuda(), delay_scale=1.0, tokens_per_step=100000)
    B, U = a.B, a.U
    src = torch.randn(B, a.samples).to(BF).cuda()
    dec = torch.randn(B, U, 256).to(BF).cuda().requires_grad_(True)
    tgt = torch.randint(2, a.V, (B, U - 1), dtype=torch.int32).cuda()
    tlen = torch.tensor(([U - 1, U * 5 // 8, U // 2, U - 8, U // 3, U // 2, U - 1, U * 2 // 3] * B)[:B], dtype=torch.int32).cuda()
    params = list(enc.parameters()) + list(proj.parameters()) + list(jn.parameters()) + list(head.parameters())

    def step(parts=None):
        for p in params:
            p.grad = None
        dec.grad = None
        t = [time.perf_counter()]
        out = enc(src, None)
        x = proj(out["encoder_out"][0])
        eo = {"encoder_out": [x], "encoder_padding_mask": [out["encoder_padding_mask"][0]]}
        joint, glen = jn(eo, dec)
        info = head.train_step(joint, tgt, glen.int(), tlen)
        return info, joint.shape

    for _ in range(3):
        info, shp = step()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        info, shp = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    rep = {"workload": "config-5 shape: base encoder twin + encoder_proj + 6-layer CAAT joiner + TransducerOut, fwd + bwd",
           "shape": {"B": B, "samples": a.samples, "joint": list(shp), "V": a.V},
           "ms_per_step": round(ms, 3), "audio_s_per_s": round(B * a.samples / 16000 / (ms / 1e3), 1), "loss": float(info["loss"])}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    return rep


if __name__ == "__main__":
    main()
