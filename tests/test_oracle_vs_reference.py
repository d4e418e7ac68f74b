"""Live comparison oracle <-> the real reference (only where /root/reference exists, i.e.
in the build container; skipped on the GPU box).  Full-width base model (d=768, 12 layers,
conv dim 512) on a short waveform, so the full-size code paths (post_extract_proj,
layer_norm_num=1, 320x2 codebook, 100 negatives) are pinned too."""
import random
import warnings

import numpy as np
import pytest
import torch

import ref_import
import w2vs_oracle as O

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not present")


@pytest.mark.parametrize("train", [False, True])
def test_base_model_matches_reference(train):
    warnings.simplefilter("ignore")
    ref = ref_import.load()
    w2 = ref.wav2vec2
    cfg = ref_import.make_cfg(ref, context_type="constant", encoder_layerdrop=0.0, dropout=0.0,
                              attention_dropout=0.0, dropout_input=0.0, dropout_features=0.0)
    torch.manual_seed(1)
    np.random.seed(1)
    random.seed(1)
    model = ref.Wav2VecSModel(cfg)
    model.train(train)
    source = torch.randn(2, 24000)
    rec = {}
    orig = w2.compute_mask_indices
    w2.compute_mask_indices = lambda *a, **k: rec.setdefault("mask", orig(*a, **k))
    osn = model.sample_negatives

    def sn(y, num):
        negs, idx = osn(y, num)
        rec["neg"] = idx.clone()
        return negs, idx

    model.sample_negatives = sn
    import torch.nn.functional as F
    ogs = F.gumbel_softmax

    def gs(logits, tau=1, hard=False, eps=1e-10, dim=-1):
        st = torch.get_rng_state()
        out = ogs(logits, tau=tau, hard=hard, eps=eps, dim=dim)
        after = torch.get_rng_state()
        torch.set_rng_state(st)
        rec["noise"] = -torch.empty_like(logits).exponential_().log()
        torch.set_rng_state(after)
        return out

    F.gumbel_softmax = gs
    try:
        with torch.set_grad_enabled(train):
            net = model(source)
    finally:
        w2.compute_mask_indices = orig
        F.gumbel_softmax = ogs
    logits = model.get_logits(net).float()
    loss = F.cross_entropy(logits, model.get_targets(None, net), reduction="sum")
    ss = logits.shape[0]
    extra = model.get_extra_losses(net)
    loss = loss + 0.1 * extra[0].float() * ss + 10.0 * extra[1].float() * ss

    P = {k: v.detach().clone().requires_grad_(train and v.dtype == torch.float32)
         for k, v in model.state_dict().items()}
    ocfg = O.OracleCfg()
    out = O.forward_loss(P, source, ocfg, mask_indices=torch.from_numpy(rec["mask"]), neg_idx=rec["neg"],
                         main_context=16, right_context=8, tau=2.0, gumbel_noise=rec.get("noise"))
    assert out["sample_size"] == ss
    np.testing.assert_allclose(out["loss"].item(), loss.item(), rtol=1e-5)
    np.testing.assert_allclose(out["logits"].detach().numpy(), logits.detach().numpy(), atol=5e-4)
    np.testing.assert_allclose(out["prob_perplexity"].item(), net["prob_perplexity"].item(), rtol=1e-5)
    np.testing.assert_allclose(out["code_perplexity"].item(), net["code_perplexity"].item(), rtol=1e-5)
    if train:
        model.zero_grad()
        loss.backward()
        out["loss"].backward()
        for n, p in model.named_parameters():
            want = p.grad
            got = P[n].grad
            if want is None:
                continue
            scale = float(want.abs().max())
            err = float((got - want).abs().max())
            assert err <= 2e-3 * scale + 1e-5, (n, err, scale)


def test_base_model_with_dropouts_matches_reference():
    """The oracle's dropout-mask INJECTION against the reference in training mode with the yaml's dropouts on
    (dropout_input / dropout_features / dropout / attention_dropout = 0.1, wav2vec-S_base_librispeech.yaml:50-77) and a
    LayerDrop-ped layer: every decision the reference draws (nn.Dropout / F.dropout call sites of wav2vec2.py:570-571,
    945-976, wav2vec_S.py:386, and the ``dropout_p`` of F.multi_head_attention_forward - reached through torch's
    scaled_dot_product_attention, which is replaced by its documented math so that the decisions are visible) is recorded as
    a multiplicative mask and handed to ``forward_loss(drop=...)``.  Pins where each mask applies (before the mask fill,
    before the right-context copies, on the softmax output, on the branch before the residual add)."""
    warnings.simplefilter("ignore")
    ref = ref_import.load()
    w2 = ref.wav2vec2
    cfg = ref_import.make_cfg(ref, context_type="constant", encoder_layerdrop=0.3, dropout=0.1, attention_dropout=0.1,
                              dropout_input=0.1, dropout_features=0.1, encoder_layers=4)
    torch.manual_seed(2)
    np.random.seed(2)
    random.seed(2)
    model = ref.Wav2VecSModel(cfg).train()
    source = torch.randn(2, 16000)
    rec = {"drops": [], "keep": []}
    orig = w2.compute_mask_indices
    w2.compute_mask_indices = lambda *a, **k: rec.setdefault("mask", orig(*a, **k))
    osn = model.sample_negatives

    def sn(y, num):
        negs, idx = osn(y, num)
        rec["neg"] = idx.clone()
        return negs, idx

    model.sample_negatives = sn
    import math
    import torch.nn.functional as F
    ogs, odrop, osdpa = F.gumbel_softmax, F.dropout, F.scaled_dot_product_attention

    def gs(logits, tau=1, hard=False, eps=1e-10, dim=-1):
        st = torch.get_rng_state()
        out = ogs(logits, tau=tau, hard=hard, eps=eps, dim=dim)
        after = torch.get_rng_state()
        torch.set_rng_state(st)
        rec["noise"] = -torch.empty_like(logits).exponential_().log()
        torch.set_rng_state(after)
        return out

    def drop(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        mask = torch.bernoulli(torch.full_like(x, 1.0 - p)) / (1.0 - p)
        rec["drops"].append(mask)
        return x * mask

    def sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, **kw):
        assert not is_causal
        s_ = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1])
        if attn_mask is not None:
            s_ = s_ + attn_mask
        return drop(torch.softmax(s_, dim=-1), dropout_p) @ v

    orand = np.random.random

    def nrand(*a):                                       # LayerDrop's draw (wav2vec_S.py:416): one per layer, after the mask
        r = orand(*a)
        if not a and "mask" in rec:
            rec["keep"].append(r > cfg.encoder_layerdrop)
        return r

    F.gumbel_softmax, F.dropout, F.scaled_dot_product_attention = gs, drop, sdpa
    np.random.random = nrand
    try:
        net = model(source)
    finally:
        w2.compute_mask_indices = orig
        F.gumbel_softmax, F.dropout, F.scaled_dot_product_attention = ogs, odrop, osdpa
        np.random.random = orand
    keep = rec["keep"]
    assert len(keep) == 4 and not all(keep) and any(keep), keep
    logits = model.get_logits(net).float()
    loss = F.cross_entropy(logits, model.get_targets(None, net), reduction="sum")
    ss = logits.shape[0]
    extra = model.get_extra_losses(net)
    loss = loss + 0.1 * extra[0].float() * ss + 10.0 * extra[1].float() * ss

    mask = torch.from_numpy(rec["mask"])
    B, T = mask.shape
    d = rec["drops"]
    assert len(d) == 3 + 3 * sum(keep), [tuple(t.shape) for t in d]
    drops = {"input": d[0], "features": d[1][mask].view(B, -1, d[1].shape[-1]), "encoder": d[2]}
    H = cfg.encoder_attention_heads
    j = 3
    for i in range(4):
        if keep[i]:
            a = d[j]
            N = a.shape[-1]
            drops[f"layer{i}"] = {"attn": a.view(B, H, N, N), "drop1": d[j + 1], "drop3": d[j + 2]}
            j += 3
    P = {k: v.detach().clone().requires_grad_(v.dtype == torch.float32) for k, v in model.state_dict().items()}
    ocfg = O.OracleCfg(encoder_layers=4)
    out = O.forward_loss(P, source, ocfg, mask_indices=mask, neg_idx=rec["neg"], main_context=16, right_context=8, tau=2.0,
                         gumbel_noise=rec["noise"], layer_keep=keep, drop=drops)
    assert out["sample_size"] == ss
    np.testing.assert_allclose(out["loss"].item(), loss.item(), rtol=1e-5)
    np.testing.assert_allclose(out["logits"].detach().numpy(), logits.detach().numpy(), atol=5e-4)
    model.zero_grad()
    loss.backward()
    out["loss"].backward()
    for n, p_ in model.named_parameters():
        want, got = p_.grad, P[n].grad
        if want is None:
            assert got is None or float(got.abs().max()) == 0, n
            continue
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-3 * scale + 1e-5, n


def test_streaming_twin_matches_reference_full_width():
    """Row f1: the reference's BlockWiseWav2Vec2Model (rain/layers/unidirect_w2v2_encoder.py) at base width on a
    padded batch, unfinished streaming call included."""
    import argparse
    warnings.simplefilter("ignore")
    rain = ref_import.load_rain()
    kw = dict(extractor_mode="layer_norm", encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072,
              encoder_attention_heads=12, final_dim=256, quantize_targets=True, feature_grad_mult=0.1, dropout=0.0,
              attention_dropout=0.0, dropout_input=0.0, dropout_features=0.0, encoder_layerdrop=0.0,
              conv_feature_layers="[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] * 2", main_context=16,
              right_context=8, pos_type="sin", load_pretrained_model_from=None)
    torch.manual_seed(3)
    model = rain.BlockWiseWav2Vec2Model.build_model(argparse.Namespace(**kw)).eval()
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    src = torch.randn(2, 20000)
    pm = torch.zeros(2, 20000, dtype=torch.bool)
    pm[1, 13000:] = True
    src[pm] = 0
    ocfg = O.OracleCfg(**{k: v for k, v in kw.items() if k in O.OracleCfg.__dataclass_fields__})
    with torch.no_grad():
        for finished, is_infer in ((False, False), (False, True)):
            r = model(src, pm, None, finished, is_infer)
            x, pad = O.streaming_encoder_forward(P, src, ocfg, main_context=16, right_context=8, padding_mask=pm,
                                                 finished=finished, is_infer=is_infer)
            assert torch.equal(pad, r["encoder_padding_mask"][0])
            valid = ~pad.transpose(0, 1)
            np.testing.assert_allclose(x[valid].numpy(), r["encoder_out"][0][valid].numpy(), atol=5e-4)
