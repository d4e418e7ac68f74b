"""Row f1 host logic that needs no GPU: argument defaults, checkpoint key surface, padding-mask helpers,
beam reordering, the API-compat mask helper, and the loud failure without a GPU."""
import argparse
import ast
import os

import numpy as np
import pytest
import torch

import w2vs_oracle as O


def test_base_architecture_defaults_and_overrides():
    from wav2vec_s_amd import streaming
    a = streaming.base_architecture(argparse.Namespace(encoder_layers=3, main_context=32))
    # rain/layers/unidirect_w2v2_encoder.py:679-745
    assert (a.encoder_layers, a.main_context, a.right_context) == (3, 32, 4)
    assert a.encoder_embed_dim == 768 and a.encoder_ffn_embed_dim == 3072 and a.encoder_attention_heads == 12
    assert a.extractor_mode == "default" and a.feature_grad_mult == 1.0 and a.required_seq_len_multiple == 2
    assert a.latent_temp == "(2,0.5,0.999995)" and a.quantize_targets is False and a.final_dim == 0
    assert eval(a.conv_feature_layers) == [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2


def test_twin_state_dict_keys_equal_reference_checkpoint(golden_dir):
    from wav2vec_s_amd import streaming
    z = np.load(os.path.join(golden_dir, "stream_twin.npz"))
    over = ast.literal_eval(bytes(z["cfg_json"]).decode())
    model = streaming.BlockWiseWav2Vec2Model.build_model(argparse.Namespace(**over))
    want = {k[len("param."):]: z[k].shape for k in z.files if k.startswith("param.")}
    got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert got == want
    assert model.cfg.context_type == "constant"
    with pytest.raises(Exception, match="MI355X|cuda"):       # no CPU path
        model(torch.zeros(1, 4000))


def test_lengths_to_padding_mask_and_frame_mask():
    from wav2vec_s_amd import streaming
    lens = torch.tensor([7, 3, 5])
    m = streaming.lengths_to_padding_mask(lens)
    assert m.shape == (3, 7) and m.dtype == torch.bool
    assert m.sum(1).tolist() == [0, 4, 2] and bool(m[1, 3]) and not bool(m[1, 2])
    # sample-level -> frame-level: a frame is padding only if ALL its samples are (wav2vec2.py:560-565)
    pm = torch.zeros(2, 1003, dtype=torch.bool)
    pm[1, 505:] = True
    f = O.frame_padding_mask(pm, 10)
    assert f.shape == (2, 10) and f[1].tolist() == [False] * 6 + [True] * 4


def test_reorder_encoder_out_matches_reference_golden(golden_dir):
    from wav2vec_s_amd import streaming
    z = np.load(os.path.join(golden_dir, "stream_online.npz"))
    enc_out = {"encoder_out": [torch.tensor(z["infer.x"])], "encoder_padding_mask": [torch.tensor(z["infer.pad"])],
               "encoder_embedding": [], "encoder_states": [], "src_tokens": [], "src_lengths": [], "dec1_state": [],
               "dec1_padding_mask": []}
    ro = streaming.OnlineW2V2TransformerEncoder.reorder_encoder_out(None, enc_out, torch.tensor(z["reorder.order"]))
    assert np.array_equal(ro["encoder_out"][0].numpy(), z["reorder.x"])
    assert np.array_equal(ro["encoder_padding_mask"][0].numpy(), z["reorder.pad"])
    assert all(ro[k] == [] for k in ("encoder_embedding", "encoder_states", "src_tokens", "src_lengths", "dec1_state",
                                     "dec1_padding_mask"))


@pytest.mark.parametrize("T,m,r", [(34, 8, 4), (50, 16, 8), (10, 16, 8), (40, 4, 0)])
def test_gen_block_atten_mask_api_helper(T, m, r):
    """The reference-spelled helper equals the oracle's block structure (itself pinned to the reference, G3)."""
    from wav2vec_s_amd import streaming
    x = torch.randn(T, 2, 8)
    pad = torch.zeros(2, T, dtype=torch.bool)
    pad[1, T - 3:] = True
    x2, pad2, attn = streaming.gen_block_atten_mask(x, pad, m, r, attn_mask_value=-1e8)
    rc_idx, rc_oob, masked = O.block_structure(T, m, r)
    n_rc = 0 if r == 0 else len(rc_idx)
    assert x2.shape[0] == T + n_rc and pad2.shape == (2, T + n_rc) and attn.shape == (T + n_rc, T + n_rc)
    assert torch.equal(attn != 0, masked) and float(attn.min()) == (-1e8 if bool(masked.any()) else 0.0)
    if r > 0:
        assert torch.equal(x2[T:], x.index_select(0, rc_idx))
        assert torch.equal(pad2[:, T:], pad.index_select(1, rc_idx) | rc_oob.unsqueeze(0))


def test_online_encoder_loads_both_checkpoint_styles(golden_dir, tmp_path):
    """rain/layers/unidirect_w2v2_encoder.py:541-556: new checkpoints carry cfg["model"], old ones an argparse Namespace in
    ckpt["args"] (for which extractor_mode / pos_type are forced); the caller's contexts replace the checkpoint's."""
    from wav2vec_s_amd import streaming
    z = np.load(os.path.join(golden_dir, "stream_online.npz"))
    over = ast.literal_eval(bytes(z["cfg_json"]).decode())
    sd = {k[len("param.w2v2_model."):]: torch.tensor(z[k]) for k in z.files if k.startswith("param.w2v2_model.")}
    new_style = os.path.join(tmp_path, "new.pt")
    old_style = os.path.join(tmp_path, "old.pt")
    torch.save({"args": None, "cfg": {"model": dict(over)}, "model": sd}, new_style)
    old_args = argparse.Namespace(**dict(over, extractor_mode="default", pos_type="conv"))
    torch.save({"args": old_args, "model": sd}, old_style)
    for path in (new_style, old_style):
        args = argparse.Namespace(w2v2_model_path=path, main_context=4, right_context=2, use_linear_layer=True,
                                  encoder_embed_dim=over["encoder_embed_dim"], freeze_finetune_updates=3)
        enc = streaming.OnlineW2V2TransformerEncoder(args)
        assert enc.encoder_proj is None                          # same width: no projection (:566-568)
        assert (enc.w2v2_model.cfg.main_context, enc.w2v2_model.cfg.right_context) == (4, 2)
        assert enc.w2v2_model.cfg.extractor_mode == "layer_norm" and enc.w2v2_model.cfg.pos_type == "sin"
        assert enc.init_frames == 6 and enc.step_frames == 4
        for k, v in enc.w2v2_model.state_dict().items():
            assert torch.equal(v, sd[k]), k
        enc.set_num_updates(2)
        assert enc.num_updates == 2 and not (enc.freeze_finetune_updates <= enc.num_updates)
    args = argparse.Namespace(w2v2_model_path=new_style, main_context=4, right_context=2, use_linear_layer=True,
                              encoder_embed_dim=48)
    enc = streaming.OnlineW2V2TransformerEncoder(args)
    assert enc.encoder_proj is not None and tuple(enc.encoder_proj.weight.shape) == (48, over["encoder_embed_dim"])
    assert enc.freeze_finetune_updates == -1                     # default: tuned from the first update
