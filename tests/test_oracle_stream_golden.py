"""Row f1 (streaming / fine-tune encoder twin): the oracle restatement
(oracle/w2vs_oracle.py: streaming_encoder_forward, online_encoder_forward) against the golden vectors
tests/golden/gen_golden_stream.py recorded from the REAL reference
(rain/layers/unidirect_w2v2_encoder.py).  CPU only."""
import ast
import os

import numpy as np
import torch

import w2vs_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _cfg(z):
    over = ast.literal_eval(bytes(z["cfg_json"]).decode())
    return O.OracleCfg(**{k: v for k, v in over.items() if k in O.OracleCfg.__dataclass_fields__}), over


def _params(z, grad):
    return {k[len("param."):]: torch.tensor(z[k]).requires_grad_(grad and z[k].dtype == np.float32)
            for k in z.files if k.startswith("param.")}


def _functional(x, pad, w):
    valid = (~pad).transpose(0, 1).unsqueeze(-1).to(x.dtype)
    return (x * torch.tensor(w) * valid).sum()


def _check_grads(z, P, prefix="grad.", has_prefix="hasgrad."):
    n_checked = 0
    for k in z.files:
        if not k.startswith(prefix):
            continue
        n = k[len(prefix):]
        want, has = z[k], bool(z[has_prefix + n][0])
        got = P[n].grad
        if not has:
            assert got is None or float(got.abs().max()) == 0.0, n
            continue
        assert got is not None, n
        scale = float(np.abs(want).max())
        err = float(np.abs(got.numpy() - want).max())
        assert err <= 2e-3 * scale + 1e-5, (n, err, scale)
        n_checked += 1
    return n_checked


def test_twin_forward_backward(golden_dir):
    z = _load(golden_dir, "stream_twin.npz")
    cfg, over = _cfg(z)
    P = _params(z, True)
    src, pm = torch.tensor(z["source"]), torch.tensor(z["padding_mask"])
    x, pad = O.streaming_encoder_forward(P, src, cfg, main_context=over["main_context"],
                                         right_context=over["right_context"], padding_mask=pm)
    assert x.shape == z["x_full"].shape                      # T x B x C
    assert np.array_equal(pad.numpy(), z["pad_full"])
    valid = ~pad.transpose(0, 1)
    np.testing.assert_allclose(x.detach()[valid].numpy(), z["x_full"][valid.numpy()], atol=2e-4)
    loss = _functional(x, pad, z["w"])
    np.testing.assert_allclose(loss.item(), z["loss"][0], rtol=1e-4, atol=1e-4)
    loss.backward()
    assert _check_grads(z, P) > 50
    # the pre-training heads take no gradient on this path
    assert not bool(z["hasgrad.final_proj.weight"][0]) and not bool(z["hasgrad.quantizer.vars"][0])
    assert not bool(z["hasgrad.mask_emb"][0])


def test_twin_streaming_trim_and_prefix(golden_dir):
    z = _load(golden_dir, "stream_twin.npz")
    cfg, over = _cfg(z)
    P = _params(z, False)
    m, r = over["main_context"], over["right_context"]
    src, pm = torch.tensor(z["source"]), torch.tensor(z["padding_mask"])
    for tag, kw in (("infer", dict(padding_mask=pm, is_infer=True)),
                    ("finished", dict(padding_mask=pm, is_infer=True, finished=True))):
        x, pad = O.streaming_encoder_forward(P, src, cfg, main_context=m, right_context=r, **kw)
        assert x.shape == z["x_" + tag].shape and np.array_equal(pad.numpy(), z["pad_" + tag]), tag
        valid = ~pad.transpose(0, 1)
        np.testing.assert_allclose(x[valid].numpy(), z["x_" + tag][valid.numpy()], atol=2e-4)
    assert z["x_infer"].shape[0] == z["x_finished"].shape[0] - r
    for tag, kw in (("prefix", {}), ("prefix_infer", dict(is_infer=True))):
        x, pad = O.streaming_encoder_forward(P, src[:, :9000], cfg, main_context=m, right_context=r, **kw)
        assert np.array_equal(pad.numpy(), z["pad_" + tag]) and not pad.any()
        np.testing.assert_allclose(x.numpy(), z["x_" + tag], atol=2e-4)
    assert z["x_prefix"].shape[0] % 2 == 1                   # odd T: the pad-to-multiple frame is removed again


def test_online_encoder_frozen_and_tuned(golden_dir):
    z = _load(golden_dir, "stream_online.npz")
    cfg, over = _cfg(z)
    args = ast.literal_eval(bytes(z["args_json"]).decode())
    src, lens = torch.tensor(z["source"]), torch.tensor(z["src_lengths"])
    assert int(z["init_frames"][0]) == args["main_context"] + args["right_context"]
    assert int(z["step_frames"][0]) == args["main_context"]
    for tag in ("frozen", "tuned"):
        P = _params(z, True)
        if tag == "frozen":                                   # num_updates < freeze_finetune_updates: no_grad twin
            for k, v in P.items():
                if k.startswith("w2v2_model."):
                    v.requires_grad_(False)
        x, pad = O.online_encoder_forward(P, src, lens, cfg, main_context=args["main_context"],
                                          right_context=args["right_context"])
        assert np.array_equal(pad.numpy(), z[tag + ".pad"])
        valid = ~pad.transpose(0, 1)
        np.testing.assert_allclose(x.detach()[valid].numpy(), z[tag + ".x"][valid.numpy()], atol=2e-4)
        loss = _functional(x, pad, z["w"])
        np.testing.assert_allclose(loss.item(), z[tag + ".loss"][0], rtol=1e-4, atol=1e-4)
        loss.backward()
        n = _check_grads(z, P, prefix=tag + ".grad.", has_prefix=tag + ".hasgrad.")
        assert n == 2 if tag == "frozen" else n > 40          # frozen: encoder_proj.{weight,bias} only
    x, pad = O.online_encoder_forward(_params(z, False), src, lens, cfg, main_context=args["main_context"],
                                      right_context=args["right_context"], is_infer=True)
    assert np.array_equal(pad.numpy(), z["infer.pad"])
    valid = ~pad.transpose(0, 1)
    np.testing.assert_allclose(x[valid].numpy(), z["infer.x"][valid.numpy()], atol=2e-4)
