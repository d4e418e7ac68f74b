"""Host logic of the training step (no GPU): LR schedule known answers, flat parameter storage round trips."""
import numpy as np
import pytest
import torch

import w2vs_oracle as O


def test_polynomial_decay_known_answers():
    """fs/optim/lr_scheduler/polynomial_decay_schedule.py:74-89 with the base yaml's numbers (lr 5e-4, warm-up 5 000,
    max_update 400 000; wav2vec-S_base_librispeech.yaml:37-48).  Expected values are the formula evaluated by hand."""
    from wav2vec_s_amd.trainer import PolynomialDecayLRSchedule
    s = PolynomialDecayLRSchedule([5e-4], warmup_updates=5000, total_num_update=400000)
    assert s.current == pytest.approx(5e-4 / 5000)          # __init__: warmup_factor * lr (:50-57)
    want = {0: 0.0, 1: 1e-7, 2500: 2.5e-4, 5000: 5e-4, 200000: 5e-4 * (1 - 195000 / 395000), 399999: 5e-4 / 395000,
            400000: 0.0, 500000: 0.0}
    for n, v in want.items():
        assert s.step_update(n) == pytest.approx(v, rel=1e-12, abs=1e-18), n
        assert O.polynomial_decay_lr(n, 5e-4, 5000, 400000) == pytest.approx(v, rel=1e-12, abs=1e-18)
    # no warm-up, power 2, non-zero floor
    s2 = PolynomialDecayLRSchedule(1e-3, warmup_updates=0, total_num_update=100, end_learning_rate=1e-5, power=2.0)
    for n in (0, 1, 50, 99, 100):
        assert s2.step_update(n) == pytest.approx(O.polynomial_decay_lr(n, 1e-3, 0, 100, 1e-5, 2.0), rel=1e-12)
    assert s2.step_update(50) == pytest.approx((1e-3 - 1e-5) * 0.25 + 1e-5)


def test_clip_coef_restatement():
    assert O.clip_coef(10.0, 0.0) == 1.0
    assert O.clip_coef(10.0, 25.0) == 1.0
    assert O.clip_coef(50.0, 25.0) == pytest.approx(25.0 / (50.0 + 1e-6))


def test_flat_params_master_sync_and_state_dict_cpu():
    """FlatParams is pure tensor plumbing: parameters become views of one bf16 buffer, the fp32 master follows a
    load_state_dict, optimizer state round-trips, a foreign layout is refused."""
    import wav2vec_s_amd as w
    from wav2vec_s_amd import trainer
    kw = dict(quantize_targets=True, extractor_mode="layer_norm", final_dim=32, encoder_embed_dim=32,
              encoder_ffn_embed_dim=64, encoder_attention_heads=2, encoder_layers=2, latent_vars=8, num_negatives=5,
              conv_feature_layers="[(16, 10, 5)] + [(16, 3, 2)] * 4 + [(16,2,2)] * 2")
    torch.manual_seed(0)
    donor = w.Wav2VecSModel(w.Wav2VecSConfig(**kw)).to(torch.bfloat16)
    sd = {k: v.clone() for k, v in donor.state_dict().items()}
    torch.manual_seed(1)
    model = w.Wav2VecSModel(w.Wav2VecSConfig(**kw)).to(torch.bfloat16)
    flat = trainer.FlatParams(model)
    assert torch.equal(flat.p32, flat.p16.float())
    before = flat.p32.clone()
    model.load_state_dict(sd)
    assert not torch.equal(flat.p32, before)
    assert torch.equal(flat.p32, flat.p16.float())                    # the hook re-derived the master
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k                               # and the reference layout is what state_dict shows
    with torch.no_grad():
        model.mask_emb.data.fill_(0.5)
    assert float(flat.p32[flat.arena.offsets["mask_emb"][0]]) != 0.5  # manual writes need the explicit call
    flat.sync_master_from_model()
    assert float(flat.p32[flat.arena.offsets["mask_emb"][0]]) == 0.5
    flat.step, flat.m[:] = 7, 0.25
    osd = flat.state_dict()
    torch.manual_seed(2)
    model2 = w.Wav2VecSModel(w.Wav2VecSConfig(**kw)).to(torch.bfloat16)
    flat2 = trainer.FlatParams(model2)
    flat2.load_state_dict(osd)
    assert flat2.step == 7 and torch.equal(flat2.m, flat.m) and torch.equal(flat2.p32, flat.p32)
    assert torch.equal(model2.mask_emb.data.float(), model.mask_emb.data.float())
    kw3 = dict(kw, encoder_layers=3)
    flat3 = trainer.FlatParams(w.Wav2VecSModel(w.Wav2VecSConfig(**kw3)).to(torch.bfloat16))
    with pytest.raises(ValueError):
        flat3.load_state_dict(osd)
