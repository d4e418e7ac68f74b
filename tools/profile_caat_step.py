#!/usr/bin/env python3
"""Config-5-shaped step (tools/bench_caat.py): host issue time vs total, per stage.   python tools/profile_caat_step.py"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from wav2vec_s_amd import joiner, streaming, transducer  # noqa: E402

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
V, B, U = 8000, 8, 48
head = transducer.TransducerOut(torch.nn.Linear(256, V, bias=False).to(BF).cuda(), delay_scale=1.0, tokens_per_step=100000)
src = torch.randn(B, 96000).to(BF).cuda()
dec = torch.randn(B, U, 256).to(BF).cuda().requires_grad_(True)
tgt = torch.randint(2, V, (B, U - 1), dtype=torch.int32).cuda()
tlen = torch.tensor([47, 30, 24, 40, 16, 24, 47, 32], dtype=torch.int32).cuda()
params = list(enc.parameters()) + list(proj.parameters()) + list(jn.parameters()) + list(head.parameters())


def stage_times(sync):
    for p in params:
        p.grad = None
    dec.grad = None
    t = [time.perf_counter()]

    def mark():
        if sync:
            torch.cuda.synchronize()
        t.append(time.perf_counter())
    out = enc(src, None); mark()
    x = proj(out["encoder_out"][0])
    joint, glen = jn({"encoder_out": [x], "encoder_padding_mask": [out["encoder_padding_mask"][0]]}, dec); mark()
    info = head.train_step(joint, tgt, glen.int(), tlen); mark()
    torch.cuda.synchronize(); t.append(time.perf_counter())
    return [(b - a) * 1e3 for a, b in zip(t, t[1:])]


for _ in range(3):
    stage_times(False)
for sync in (True, False):
    acc = None
    for _ in range(5):
        r = stage_times(sync)
        acc = r if acc is None else [a + b for a, b in zip(acc, r)]
    print(("synchronised stages" if sync else "issue only        "), "encoder fwd %.2f | proj+joiner fwd %.2f | head step incl. ALL backward %.2f | drain %.2f  (ms)" % tuple(a / 5 for a in acc), flush=True)

if os.environ.get("W2VS_CPROFILE") == "1":
    import cProfile
    import pstats
    torch.autograd.set_multithreading_enabled(False)      # the backward nodes run on THIS thread: cProfile sees them
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        stage_times(False)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
