"""BASELINE configs[4] (streaming-ST fine-tune) as a synthetic step, at two shapes:

  "script"  what wav2vec_s_scripts/train/train_wav2vec_s_caat_simulst_base.sh:17-41 trains: the joiner at 768 / 12 heads / ffn 3072,
            6 layers, --transducer-downsample 64 --step-mode random (rain/layers/attention_transducer.py:800-808 draws the step
            from {2, 4, 10, 20} x 16 frames per forward), dropout 0.3, activation-dropout 0.1, attention-dropout 0.1,
            tokens-per-step 6000, delay_func diag_positive, delay_scale 1, label smoothing 0.1.  encoder_embed_dim 768 equals the
            wav2vec-S width, so --use-linear-layer builds no encoder_proj (rain/layers/unidirect_w2v2_encoder.py:559-561).
            MuST-C en-de-sized batch: max-tokens 1 400 000 samples = 14 utterances of 100 000 samples (6.25 s, 312 frames), 31
            target tokens (+ 1), a 10 000-piece vocabulary.
  "arch"    the `w2v2_caat` architecture defaults (rain/models/w2v2_transducer.py:334-340): 256 / 4 / 1024, downsample 16,
            constant step, encoder_proj 768 -> 256 - what rounds 2-4 measured.

The autoregressive text decoder that produces the decoder states (a fairseq TransformerDecoder, outside SURVEY section 8) is
replaced by a random-state input; its cost is not measured."""
import argparse

import torch

BF = torch.bfloat16

SHAPES = {
    "script": dict(D=768, H=12, ffn=3072, ds=64, step_mode="random", dropout=0.3, act_dropout=0.1, attn_dropout=0.1,
                   tokens_per_step=6000, delay_func="diag_positive", B=14, samples=100000, U=32, V=10000),
    "arch": dict(D=256, H=4, ffn=1024, ds=16, step_mode="constant", dropout=0.1, act_dropout=0.1, attn_dropout=0.1,
                 tokens_per_step=100000, delay_func="zero", B=8, samples=96000, U=48, V=8000),
}

ENC_KW = dict(extractor_mode="layer_norm", encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072,
              encoder_attention_heads=12, final_dim=256, quantize_targets=True, feature_grad_mult=0.1, dropout=0.1,
              attention_dropout=0.1, dropout_input=0.0, dropout_features=0.0, encoder_layerdrop=0.0,
              conv_feature_layers="[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] * 2", main_context=16, right_context=8,
              pos_type="sin", load_pretrained_model_from=None)


def joiner_args(s, train_dropouts=True):
    p = 1.0 if train_dropouts else 0.0
    return argparse.Namespace(jointer_embed_dim=s["D"], jointer_attention_heads=s["H"], transducer_downsample=s["ds"],
                              jointer_layers=6, attention_dropout=s["attn_dropout"] * p, dropout=s["dropout"] * p,
                              activation_dropout=s["act_dropout"] * p, activation_fn="relu", encoder_normalize_before=True,
                              jointer_ffn_embed_dim=s["ffn"], step_mode=s["step_mode"])


def build(shape, B=None, samples=None, U=None, V=None, seed=0):
    """-> dict(step=callable returning (info, joint shape), params, shape=dict)."""
    from wav2vec_s_amd import joiner, streaming, transducer
    s = dict(SHAPES[shape])
    for k, v in (("B", B), ("samples", samples), ("U", U), ("V", V)):
        if v:
            s[k] = v
    torch.manual_seed(seed)
    enc = streaming.BlockWiseWav2Vec2Model.build_model(argparse.Namespace(**ENC_KW)).to(BF).cuda().train()
    proj = streaming.HipLinear(768, s["D"]).to(BF).cuda() if s["D"] != 768 else None
    jn = joiner.MHAJointNet(joiner_args(s)).to(BF).cuda().train()
    head = transducer.TransducerOut(torch.nn.Linear(s["D"], s["V"], bias=False).to(BF).cuda(), delay_scale=1.0,
                                    tokens_per_step=s["tokens_per_step"], label_smoothing=0.1, delay_func=s["delay_func"])
    Bn, Un = s["B"], s["U"]
    src = torch.randn(Bn, s["samples"]).to(BF).cuda()
    dec = torch.randn(Bn, Un, s["D"]).to(BF).cuda().requires_grad_(True)
    tgt = torch.randint(2, s["V"], (Bn, Un - 1), dtype=torch.int32).cuda()
    tlen = torch.tensor(([Un - 1, Un * 5 // 8, Un // 2, max(Un - 8, 1), Un // 3, Un // 2, Un - 1, Un * 2 // 3] * Bn)[:Bn],
                        dtype=torch.int32).cuda()
    params = list(enc.parameters()) + (list(proj.parameters()) if proj is not None else []) + list(jn.parameters()) \
        + list(head.parameters())

    def stages(mark=lambda: None):
        for p in params:
            p.grad = None
        dec.grad = None
        out = enc(src, None)
        mark()
        x = out["encoder_out"][0]
        if proj is not None:
            x = proj(x)
        joint, glen = jn({"encoder_out": [x], "encoder_padding_mask": [out["encoder_padding_mask"][0]]}, dec)
        mark()
        info = head.train_step(joint, tgt, glen.int(), tlen)
        mark()
        return info, tuple(joint.shape)

    return dict(step=stages, params=params, shape=s, jn=jn, enc=enc, head=head)
