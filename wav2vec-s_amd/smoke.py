"""One small invocation of the hot path on cuda:0, checked against the CPU oracle
(used by __graft_entry__.smoke).  Tiny base-style model (2 layers, d=128, conv dim 64): forward +
InfoNCE/diversity/penalty loss + backward, with every host draw injected into both sides."""
import numpy as np
import torch


def run():
    import w2vs_oracle as O   # checker only (smoke is allowed to use the oracle)
    from . import Wav2VecSConfig, engine, host_rng
    from .criterion import Wav2vecCriterion
    from .model import Wav2VecSModel
    kw = dict(quantize_targets=True, extractor_mode="layer_norm", final_dim=128, encoder_layers=2, encoder_embed_dim=128,
              encoder_ffn_embed_dim=256, encoder_attention_heads=2, encoder_layerdrop=0.0, dropout=0.0,
              attention_dropout=0.0, dropout_input=0.0, dropout_features=0.0, feature_grad_mult=0.1, latent_vars=40,
              num_negatives=20, context_type="constant",
              conv_feature_layers="[(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64,2,2)] * 2")
    cfg = Wav2VecSConfig(**kw)
    torch.manual_seed(0)
    np.random.seed(0)
    model = Wav2VecSModel(cfg).to(torch.bfloat16)
    P = {k: v.float().clone().requires_grad_(v.dtype == torch.bfloat16) for k, v in model.state_dict().items()}
    B, L = 2, 16000
    source = torch.randn(B, L).to(torch.bfloat16)
    ocfg = O.OracleCfg(**{k: v for k, v in kw.items() if k in O.OracleCfg.__dataclass_fields__})
    T = O.conv_out_lengths(L, ocfg.conv_layers)[-1]
    mask = host_rng.compute_mask_indices((B, T), None, cfg.mask_prob, cfg.mask_length, "static", 0, min_masks=2)
    M = int(mask[0].sum())
    neg = host_rng.sample_negative_indices(B, M, cfg.num_negatives)
    noise = -torch.empty(B * M * cfg.latent_groups, cfg.latent_vars).exponential_().log()
    model = model.cuda().train()
    model.inject_draws(engine.Draws(mask_indices=mask, neg_idx=neg, context=(8, 4), layer_keep=[True, True],
                                    gumbel_noise=noise))
    crit = Wav2vecCriterion(infonce=True, loss_weights=[0.1, 10.0])
    loss, sample_size, _ = crit(model, {"net_input": {"source": source.cuda()}})
    loss.backward()
    st = model._last_state
    ref = O.forward_loss(P, source.float(), ocfg, mask_indices=torch.from_numpy(mask), neg_idx=neg, main_context=8,
                         right_context=4, tau=2.0, gumbel_noise=noise, force_code_idx=st.qst.idx.cpu())
    ref["loss"].backward()
    rel = abs(float(loss) - float(ref["loss"])) / abs(float(ref["loss"]))
    assert sample_size == ref["sample_size"], (sample_size, ref["sample_size"])
    assert rel < 2e-3, (float(loss), float(ref["loss"]))
    g = model.encoder.layers[0].fc1.weight.grad.float().cpu()
    w = P["encoder.layers.0.fc1.weight"].grad
    gerr = float((g - w).norm() / w.norm())
    assert gerr < 0.1, gerr
    print("smoke ok: loss %.4f vs oracle %.4f (rel %.1e), fc1 grad rel err %.3f" % (float(loss), float(ref["loss"]), rel, gerr))
