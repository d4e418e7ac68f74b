"""Callable module surface of the reference classes (VERDICT r1 'missing' #5): the helpers other code
calls on the model / its sub-modules, on the HIP kernels, against the oracle."""
import numpy as np
import pytest
import torch

import w2vs_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16

KW = dict(quantize_targets=True, extractor_mode="layer_norm", final_dim=128, encoder_layers=2, encoder_embed_dim=128,
          encoder_ffn_embed_dim=256, encoder_attention_heads=2, latent_vars=40, num_negatives=20,
          conv_feature_layers="[(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64,2,2)] * 2")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _model(**over):
    import wav2vec_s_amd as w
    torch.manual_seed(0)
    m = w.Wav2VecSModel(w.Wav2VecSConfig(**dict(KW, **over))).to(BF)
    P = {k: v.float() for k, v in m.state_dict().items()}
    return m.cuda().eval(), P, O.OracleCfg(**{k: v for k, v in dict(KW, **over).items() if k in O.OracleCfg.__dataclass_fields__})


@pytest.mark.parametrize("mode", ["layer_norm", "default"])
def test_conv_feature_extractor_forward(mode):
    m, P, ocfg = _model(extractor_mode=mode)
    src = torch.randn(2, 8000, generator=torch.Generator().manual_seed(1)).to(BF)
    got = m.feature_extractor(src.cuda())                               # B x C x T'
    want = O.conv_feature_extractor(src.float(), P, ocfg)
    assert tuple(got.shape) == tuple(want.shape)
    assert rel(got, want) < 1.5e-2


def test_quantize_and_sample_negatives():
    m, P, ocfg = _model()
    src = torch.randn(2, 8000, generator=torch.Generator().manual_seed(2)).to(BF)
    q, idx = m.quantize(src.cuda())
    feats = O.conv_feature_extractor(src.float(), P, ocfg).transpose(1, 2)
    feats = torch.nn.functional.layer_norm(feats, (64,), P["layer_norm.weight"], P["layer_norm.bias"], 1e-5)
    qr, idxr, _, _ = O.gumbel_quantize(feats, P, ocfg, 1.0, None)
    B, T = feats.shape[:2]
    assert tuple(q.shape) == (B, T, qr.shape[-1]) and tuple(idx.shape) == (B, T, ocfg.latent_groups)
    same = (idx.cpu().view(-1, ocfg.latent_groups) == idxr).all(-1)
    # near-ties of the bf16 features may flip (48 rows: one flip is 2 %): every disagreement must BE a near-tie in the oracle's
    # own logits - margin between its best code and the HIP choice below 5 % of the logit spread (~5 steps of the features' bf16 noise) - as tests/test_model_gpu.py asks
    G, V = ocfg.latent_groups, ocfg.latent_vars
    ql = torch.nn.functional.linear(feats.reshape(-1, feats.shape[-1]), P["quantizer.weight_proj.weight"],
                                    P["quantizer.weight_proj.bias"]).view(-1, V)
    hv = ql.gather(1, idx.cpu().long().view(-1, 1)).view(-1)
    assert float((ql.max(-1).values - hv).max()) < 0.05 * float(ql.std())
    assert float(same.float().mean()) > 0.9
    assert rel(q.view(B * T, -1)[same.cuda()], qr.view(B * T, -1)[same]) < 1e-2
    # sample_negatives: same torch.randint draws as the reference helper, rows gathered on the GPU
    y = torch.randn(2, 31, 128).to(BF).cuda()
    torch.manual_seed(11)
    negs, nidx = m.sample_negatives(y, 31)
    torch.manual_seed(11)
    want_idx = O.sample_negative_indices(2, 31, 20)
    assert torch.equal(nidx, want_idx) and tuple(negs.shape) == (20, 2, 31, 128)
    want = y.view(-1, 128)[want_idx.view(-1).cuda()].view(2, 31, 20, 128).permute(2, 0, 1, 3)
    assert torch.equal(negs, want)


@pytest.mark.parametrize("pre_ln", [False, True])
def test_encoder_layer_forward(pre_ln):
    m, P, ocfg = _model(layer_norm_first=pre_ln)
    layer = m.encoder.layers[1]
    T, B, E = 37, 3, 128
    x = torch.randn(T, B, E, generator=torch.Generator().manual_seed(3)).to(BF)
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[2, 30:] = True
    y, attn = layer(x.cuda(), self_attn_padding_mask=pad.cuda())
    add = torch.zeros(B, 1, 1, T).masked_fill(pad.view(B, 1, 1, T), float("-inf")).expand(B, 1, T, T)
    want = O.encoder_layer(x.float(), P, "encoder.layers.1.", ocfg, add)
    assert attn is None and tuple(y.shape) == (T, B, E)
    valid = (~pad).transpose(0, 1)
    assert rel(y[valid.cuda()], want[valid]) < 1.5e-2
    with pytest.raises(Exception):
        layer(x.cuda(), self_attn_mask=torch.zeros(T, T).cuda())
