"""The oracle (oracle/w2vs_oracle.py) against the committed golden vectors that
tests/golden/gen_golden.py recorded from the REAL reference.  CPU only."""
import ast
import hashlib
import os

import numpy as np
import pytest
import torch

import w2vs_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _cfg_from(z):
    over = ast.literal_eval(bytes(z["cfg_json"]).decode())
    keep = {k: v for k, v in over.items() if k in O.OracleCfg.__dataclass_fields__}
    cfg = O.OracleCfg(**keep)
    cfg.loss_weights = tuple(float(v) for v in z["loss_weights"])
    return cfg, over


# ------------------------------------------------------------------ G1 masks
@pytest.mark.parametrize("B,T", [(2, 499), (8, 546), (5, 781), (3, 999), (2, 49)])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_mask_indices_bit_exact(golden_dir, B, T, seed):
    z = _load(golden_dir, "host_rng.npz")
    np.random.seed(seed)
    m = O.compute_mask_indices((B, T), None, 0.65, 10, "static", 0, min_masks=2)
    want = np.unpackbits(z[f"mask_{B}_{T}_s{seed}"], axis=1)[:, :T].astype(bool)
    assert np.array_equal(m, want)
    # same number of draws consumed from the global RNG
    assert np.random.rand() == z[f"mask_{B}_{T}_s{seed}_next"][0]
    assert len(set(m.sum(1))) == 1  # every row has exactly M masked frames


def test_mask_indices_with_padding(golden_dir):
    z = _load(golden_dir, "host_rng.npz")
    np.random.seed(3)
    pm = torch.zeros(3, 200, dtype=torch.bool)
    pm[1, 150:] = True
    pm[2, 90:] = True
    m = O.compute_mask_indices((3, 200), pm, 0.65, 10, "static", 0, min_masks=2)
    want = np.unpackbits(z["mask_pad_3_200_s3"], axis=1)[:, :200].astype(bool)
    assert np.array_equal(m, want)


# ------------------------------------------------------------------ G2 negatives
@pytest.mark.parametrize("B,M,seed", [(2, 20, 0), (2, 247, 1), (8, 245, 2)])
def test_negative_indices_bit_exact(golden_dir, B, M, seed):
    z = _load(golden_dir, "host_rng.npz")
    torch.manual_seed(seed)
    idx = O.sample_negative_indices(B, M, 100).numpy()
    assert idx.dtype == np.int64
    sha = hashlib.sha256(np.ascontiguousarray(idx).tobytes()).digest()
    assert sha == bytes(z[f"neg_{B}_{M}_s{seed}_sha"])
    assert np.array_equal(idx[:, :32], z[f"neg_{B}_{M}_s{seed}_head"])
    # never self, always inside own utterance
    own = np.repeat(np.arange(M), 100)[None, :] + (np.arange(B) * M)[:, None]
    assert (idx != own).all()
    assert ((idx // M) == np.arange(B)[:, None]).all()


# ------------------------------------------------------------------ G3 block structure
@pytest.mark.parametrize("Tp,m,r", [(500, 16, 8), (546, 16, 8), (34, 8, 4), (40, 32, 16), (10, 16, 8),
                                    (50, 8, 0), (48, 16, 8)])
def test_block_structure(golden_dir, Tp, m, r):
    z = _load(golden_dir, "host_rng.npz")
    rc_idx, rc_oob, masked = O.block_structure(Tp, m, r)
    src = np.concatenate([np.arange(Tp), rc_idx.numpy()]).astype(np.int32)
    assert np.array_equal(src, z[f"blk_{Tp}_{m}_{r}_src"])
    N = Tp + len(rc_idx)
    want = np.unpackbits(z[f"blk_{Tp}_{m}_{r}_mask"], axis=1)[:, :N].astype(bool)
    assert np.array_equal(masked.numpy(), want)
    pad = torch.zeros(2, Tp, dtype=torch.bool)
    pad[1, Tp - 1] = True
    if r > 0:
        pad = torch.cat([pad, pad.index_select(1, rc_idx) | rc_oob.unsqueeze(0)], dim=1)
    assert np.array_equal(pad.numpy(), z[f"blk_{Tp}_{m}_{r}_pad"])
    vals = set(float(v) for v in z[f"blk_{Tp}_{m}_{r}_vals"])
    assert vals <= {0.0, -1e4}  # finite mask value, not -inf (wav2vec_S.py:487)


# ------------------------------------------------------------------ G4 sinusoid
def test_sinusoid_rows(golden_dir):
    z = _load(golden_dir, "host_rng.npz")
    t = O.sinusoidal_table(8002, 768, 1)
    assert np.array_equal(t[[0, 1, 2, 3, 500, 8001]].numpy(), z["sin768_rows"])
    pad = torch.tensor([[False, False, True, False], [False, False, False, False]])
    pos = O.positions_from_padding(pad)
    assert pos.tolist() == [[2, 3, 1, 4], [2, 3, 4, 5]]
    got = t.index_select(0, pos.view(-1)).view(2, 4, 768)[:, :, :4].numpy()
    assert np.array_equal(got, z["sin768_bool_input"])


# ------------------------------------------------------------------ G5 tiny models end to end
def _run_tiny(z, cfg, over, train=True, layer_keep=None, ctx=None):
    P = {k[len("param."):]: torch.tensor(z[k]).requires_grad_(z[k].dtype == np.float32 and train)
         for k in z.files if k.startswith("param.")}
    m = over.get("main_context", 16) if ctx is None else ctx[0]
    r = over.get("right_context", 8) if ctx is None else ctx[1]
    noise = torch.tensor(z["gumbel_noise"]) if "gumbel_noise" in z.files else None
    tau = float(z["tau"][0]) if "tau" in z.files else float(z["temp"][0])
    col = {}
    out = O.forward_loss(P, torch.tensor(z["source"]), cfg, mask_indices=torch.tensor(z["mask_indices"]),
                         neg_idx=torch.tensor(z["neg_idx"]), main_context=m, right_context=r, tau=tau,
                         gumbel_noise=noise, layer_keep=layer_keep, collect=col)
    return P, out, col


def _check_common(z, out, col):
    assert out["sample_size"] == int(z["sample_size"][0])
    np.testing.assert_allclose(out["loss"].item(), z["loss"][0], rtol=1e-5)
    np.testing.assert_allclose(out["features_pen"].item(), z["features_pen"][0], rtol=1e-5)
    np.testing.assert_allclose(out["prob_perplexity"].item(), z["prob_perplexity"][0], rtol=1e-5)
    np.testing.assert_allclose(out["code_perplexity"].item(), z["code_perplexity"][0], rtol=1e-5)
    np.testing.assert_allclose(out["logits"].detach().numpy(), z["logits"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(col["conv0"].detach().numpy(), z["act.conv0"], atol=1e-5)
    np.testing.assert_allclose(col["conv6"].detach().numpy(), z["act.conv_out"], atol=1e-5)
    np.testing.assert_allclose(col["features"].detach().numpy(), z["act.features"], atol=1e-4)
    np.testing.assert_allclose(col["enc_out"].detach().numpy(), z["act.enc_out"], atol=1e-4)
    np.testing.assert_allclose(col["q"].detach().numpy(), z["act.q"], atol=1e-5)
    np.testing.assert_allclose(col["yq"].detach().numpy(), z["act.yq"], atol=1e-5)
    np.testing.assert_allclose(col["xf"].detach().numpy(), z["act.xf"], atol=1e-4)


def _check_grads(z, P, out, skip_none=True):
    out["loss"].backward()
    worst = 0.0
    for k in z.files:
        if not k.startswith("grad."):
            continue
        n = k[len("grad."):]
        want = z[k]
        has = bool(z["hasgrad." + n][0])
        got = P[n].grad
        if not has:
            assert got is None or float(got.abs().max()) == 0.0, n
            continue
        assert got is not None, n
        # k_proj.bias has an analytically ZERO gradient (softmax is shift invariant), so it
        # holds only rounding noise: absolute floor 1e-5 next to the relative bound
        scale = float(np.abs(want).max())
        err = float(np.abs(got.numpy() - want).max())
        worst = max(worst, err / max(scale, 1e-6))
        assert err <= 2e-3 * scale + 1e-5, (n, err, scale)
    return worst


def test_tiny_base_forward_backward(golden_dir):
    z = _load(golden_dir, "tiny_base.npz")
    cfg, over = _cfg_from(z)
    P, out, col = _run_tiny(z, cfg, over)
    _check_common(z, out, col)
    np.testing.assert_allclose(col["layer0"].detach().numpy(), z["act.layer0"], atol=1e-4)
    _check_grads(z, P, out)


def test_tiny_large_style_forward_backward(golden_dir):
    z = _load(golden_dir, "tiny_large.npz")
    cfg, over = _cfg_from(z)
    assert cfg.layer_norm_first and cfg.conv_bias and cfg.layer_norm_num == 7
    P, out, col = _run_tiny(z, cfg, over)
    _check_common(z, out, col)
    _check_grads(z, P, out)


def test_tiny_eval_groupnorm_forward(golden_dir):
    z = _load(golden_dir, "tiny_eval_gn.npz")
    cfg, over = _cfg_from(z)
    assert cfg.extractor_mode == "default"
    with torch.no_grad():
        P, out, col = _run_tiny(z, cfg, over, train=False)
    _check_common(z, out, col)


def test_tiny_layerdrop_sampled_context(golden_dir):
    z = _load(golden_dir, "tiny_layerdrop.npz")
    cfg, over = _cfg_from(z)
    draws = z["layerdrop_draws"]
    assert len(draws) == cfg.encoder_layers  # one np.random.random() per layer (wav2vec_S.py:415)
    keep = [bool(d > over["encoder_layerdrop"]) for d in draws]
    assert not all(keep)
    a, b = [int(v) for v in z["context_draws"]]
    m = a * 2
    r = min(b * 2, m // 2)  # wav2vec_S.py:393-395
    P, out, col = _run_tiny(z, cfg, over, layer_keep=keep, ctx=(m, r))
    _check_common(z, out, col)
    _check_grads(z, P, out)
    for i, k in enumerate(keep):
        if not k:
            assert P[f"encoder.layers.{i}.fc1.weight"].grad is None
