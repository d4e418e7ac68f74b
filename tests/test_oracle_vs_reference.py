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
