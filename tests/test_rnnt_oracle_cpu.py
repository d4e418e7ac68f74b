"""Row f4 (transducer loss): the numpy oracle (oracle/rnnt_oracle.py) against the known answers the reference's
tests hold and against the reference's own CPU implementation compiled into oracle/_ref.  CPU only."""
import json
import os

import numpy as np
import pytest

import rnnt_oracle as R
from conftest import GOLDEN


@pytest.fixture(scope="module")
def ka(golden_dir):
    return json.load(open(os.path.join(golden_dir, "rnnt_known_answers.json")))


def _case(c):
    acts = np.array(c["acts"], dtype=np.float64).reshape(c["B"], c["T"], c["U"], c["V"])
    return acts, np.array(c["labels"]), np.array(c["input_lengths"]), np.array(c["label_lengths"])


def test_known_answers_small_and_options(ka):
    acts, lab, xl, yl = _case(ka["small"])
    costs, _ = R.rnnt_loss(acts, lab, xl, yl, blank=0)
    assert abs(costs[0] - ka["small"]["expected_score"]) < ka["small"]["tol"]
    c = ka["options"]
    acts, lab, xl, yl = _case(c)
    costs, grads = R.rnnt_loss(acts, lab, xl, yl, blank=0)
    np.testing.assert_allclose(costs, c["expected_scores"], atol=c["tol"])
    np.testing.assert_allclose(grads.reshape(-1), c["expected_grads_wrt_acts"], atol=c["tol"])
    # the delay entry point reports the same NLL in costs[0] and, with delay_scale = 0 and smooth = 1, the same gradient
    dv = R.delay_cost("zero", c["B"], c["T"], c["U"], xl, yl)
    costs3, g3 = R.delay_loss(acts, lab, xl, yl, dv, delay_scale=0.0, smooth=1.0)
    np.testing.assert_allclose(costs3[0], c["expected_scores"], atol=c["tol"])
    np.testing.assert_allclose(costs3[2], costs3[0])
    np.testing.assert_allclose(g3.reshape(-1), c["expected_grads_wrt_acts"], atol=c["tol"])


@pytest.mark.skipif(not R.RefCpuRnnt.available(), reason="oracle/_ref/libwarprnnt_cpu.so not built (make -C oracle ref)")
def test_oracle_equals_compiled_reference_cpu(ka):
    ref = R.RefCpuRnnt()
    c = ka["options"]
    acts, lab, xl, yl = _case(c)
    lp = acts + R.log_softmax_denom(acts)[..., None]
    costs, g = ref.loss_and_logprob_grads(lp, lab, xl, yl)
    np.testing.assert_allclose(costs, c["expected_scores"], atol=c["tol"])
    np.testing.assert_allclose(g.reshape(-1), c["expected_grads_wrt_log_probs"], atol=c["tol"])
    rng = np.random.RandomState(0)
    for B, T, U, V in ((3, 17, 6, 11), (2, 40, 13, 29), (4, 9, 9, 5)):
        acts = rng.randn(B, T, U, V) * 1.5
        xl = rng.randint(max(2, T // 2), T + 1, size=B); xl[0] = T
        yl = rng.randint(1, U, size=B); yl[-1] = U - 1
        lab = rng.randint(1, V, size=(B, U - 1))
        want_c, want_g = ref.loss_and_act_grads(acts, lab, xl, yl)
        got_c, got_g = R.rnnt_loss(acts, lab, xl, yl)
        np.testing.assert_allclose(got_c, want_c, rtol=1e-5)
        np.testing.assert_allclose(got_g, want_g, atol=1e-4)      # the compiled reference computes in fp32; its own tests use 1e-4
        assert float(np.abs(got_g[0, xl[0]:]).max() if xl[0] < T else 0.0) == 0.0


def test_delay_terms_invariants():
    """The delay recursion has no CPU reference; check what the algorithm implies.  With delay values that depend on
    the frame only (the default "zero" cost): forward and backward expectations agree, and with the consistent index the
    analytic gradient is the derivative of NLL + scale * expected delay."""
    rng = np.random.RandomState(1)
    B, T, U, V = 2, 7, 4, 6
    acts = rng.randn(B, T, U, V)
    xl, yl = np.array([7, 5]), np.array([3, 2])
    lab = rng.randint(1, V, size=(B, U - 1))
    dv = R.delay_cost("zero", B, T, U, xl, yl)
    col = {}
    costs, g = R.delay_loss(acts, lab, xl, yl, dv, delay_scale=0.7, smooth=1.0, consistent_delay_index=True, collect=col)
    np.testing.assert_allclose(costs[1], col["delay_expect_bwd"], rtol=1e-9)
    np.testing.assert_allclose(costs[2], costs[0] + 0.7 * costs[1])
    assert np.all(costs[1] > 0) and np.all(costs[1] < yl)             # one s / src_len in [0, 1) per emitted label
    eps = 1e-5
    for idx in [(0, 0, 0, 0), (0, 3, 1, lab[0, 1]), (1, 4, 2, 0), (1, 2, 0, 3), (0, 6, 3, 0), (1, 1, 1, lab[1, 1])]:
        ap, am = acts.copy(), acts.copy()
        ap[idx] += eps
        am[idx] -= eps
        cp, _ = R.delay_loss(ap, lab, xl, yl, dv, delay_scale=0.7, want_grad=False)
        cm, _ = R.delay_loss(am, lab, xl, yl, dv, delay_scale=0.7, want_grad=False)
        num = (cp[2].sum() - cm[2].sum()) / (2 * eps)
        assert abs(num - g[idx]) < 1e-6, (idx, num, g[idx])
    # the reference's own index (delay_values[b * maxT + t]) gives a different - not a true - gradient; both are restated
    _, g_ref = R.delay_loss(acts, lab, xl, yl, dv, delay_scale=0.7, smooth=1.0)
    assert np.abs(g_ref - g).max() > 1e-4
    # smooth (the front end's "temperature") rescales only the likelihood part
    _, g_s = R.delay_loss(acts, lab, xl, yl, dv, delay_scale=0.0, smooth=0.5)
    _, g_1 = R.delay_loss(acts, lab, xl, yl, dv, delay_scale=0.0, smooth=1.0)
    assert np.abs(g_s - g_1).max() > 1e-3 and float(np.abs(g_s[1, 5:]).max()) == 0.0


def test_delay_cost_builders():
    d = R.delay_cost("zero", 2, 5, 3, [5, 4], [2, 2])
    assert d.shape == (2, 5, 3) and np.allclose(d[1, :, 0], np.arange(5) / 4) and np.allclose(d[..., 0], d[..., 2])
    dd = R.delay_cost("diagonal", 1, 4, 3, [4], [2])
    dp = R.delay_cost("diag_positive", 1, 4, 3, [4], [2])
    want = (np.arange(1, 5)[:, None] * 0.5 - np.arange(1, 4)[None, :]) / 2
    assert np.allclose(dd[0], np.abs(want)) and np.allclose(dp[0], np.clip(want, 0, None))


def test_oracle_equals_reference_numpy_transducer():
    """The reference's own numpy oracle (warp_transducer/pytorch_binding/test/transducer_np.py), imported from where it
    lies: costs and gradients w.r.t. log-probabilities on a ragged batch."""
    import importlib.util
    path = "/root/reference/warp_transducer/pytorch_binding/test/transducer_np.py"
    if not os.path.isfile(path):
        pytest.skip("reference tree not present")
    spec = importlib.util.spec_from_file_location("ref_transducer_np", path)
    tnp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tnp)
    rng = np.random.RandomState(4)
    B, T, U, V = 3, 11, 5, 7
    acts = rng.randn(B, T, U, V)
    xl, yl = np.array([11, 8, 6]), np.array([4, 2, 3])
    lab = rng.randint(1, V, size=(B, U - 1))
    lp = acts + R.log_softmax_denom(acts)[..., None]
    want_c, want_g = tnp.transduce_batch(lp.astype(np.float32), lab, xl, yl)
    got_c, got_g = R.rnnt_loss(acts, lab, xl, yl)
    np.testing.assert_allclose(got_c, want_c, rtol=1e-5)
    # transducer_np differentiates w.r.t. log-probabilities; chain through the softmax like RefCpuRnnt.loss_and_act_grads
    want_act = want_g.astype(np.float64) - np.exp(lp) * want_g.astype(np.float64).sum(-1, keepdims=True)
    for b in range(B):
        np.testing.assert_allclose(got_g[b, :xl[b], :yl[b] + 1], want_act[b, :xl[b], :yl[b] + 1], atol=2e-5)


def test_label_smoothed_ce_and_head_against_torch_autograd():
    """The loss head's restatement (parity unpinned: rain/layers/attention_transducer.py imports the CUDA-only
    warprnnt_pytorch and the whole fairseq Transformer stack): its cross-entropy equals fairseq's formula evaluated by torch,
    and the head's gradients equal torch autograd through the same composition."""
    import torch
    rng = np.random.RandomState(8)
    R_, V = 7, 11
    logits = rng.randn(R_, V)
    tgt = rng.randint(2, V, size=R_); tgt[2] = 1
    loss, nll, g = R.label_smoothed_ce(logits, tgt, 0.1, 1)
    lt = torch.tensor(logits, requires_grad=True)
    lp = torch.log_softmax(lt, -1)
    t = torch.tensor(tgt).unsqueeze(-1)
    nl = -lp.gather(-1, t); sm = -lp.sum(-1, keepdim=True)
    m = t.eq(1); nl = nl.masked_fill(m, 0.0).sum(); sm = sm.masked_fill(m, 0.0).sum()       # label_smoothed_cross_entropy.py:36-47
    eps_i = 0.1 / (V - 1)
    want = (1.0 - 0.1 - eps_i) * nl + eps_i * sm
    want.backward()
    np.testing.assert_allclose(loss, want.item(), rtol=1e-12)
    np.testing.assert_allclose(nll, nl.item(), rtol=1e-12)
    np.testing.assert_allclose(g, lt.grad.numpy(), atol=1e-12)
    B, T, U, d, V = 2, 6, 4, 5, 9
    x, W = rng.randn(B, T, U, d), rng.randn(V, d) * 0.5
    tg = rng.randint(2, V, size=(B, U - 1)); tg[1, 2] = 1
    sl, tl = np.array([6, 4]), np.array([3, 2])
    out, dx, dW = R.transducer_out_step(x, W, tg, sl, tl, delay_scale=0.0, ce_scale=0.7, loss_scale=2.0)
    # with delay_scale = 0 the transducer term is the plain NLL, whose gradient torch can reproduce from the oracle's dlogits
    xt, Wt = torch.tensor(x, requires_grad=True), torch.tensor(W, requires_grad=True)
    logits = xt @ Wt.T
    _, dl = R.rnnt_loss(logits.detach().numpy(), tg, sl, tl)
    last = xt[torch.arange(B), torch.tensor(sl) - 1][:, :-1]
    lp = torch.log_softmax((last @ Wt.T).reshape(B * (U - 1), V), -1)
    t = torch.tensor(tg).reshape(-1, 1)
    nl = (-lp.gather(-1, t)).masked_fill(t.eq(1), 0.0).sum(); sm = (-lp.sum(-1, keepdim=True)).masked_fill(t.eq(1), 0.0).sum()
    eps_i = 0.1 / (V - 1)
    ce = (1.0 - 0.1 - eps_i) * nl + eps_i * sm
    (2.0 * ((logits * torch.tensor(dl)).sum() + 0.7 * ce)).backward()
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(dW, Wt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(out["loss"], out["loss_prob"] + 0.7 * ce.item(), rtol=1e-12)


@pytest.mark.parametrize("tag", ["mb1", "mb3", "mb3_scaled"])
def test_transducer_out_oracle_matches_reference_fixture(tag):
    """The loss head's oracle against the REFERENCE's TransducerOut.train_step (rain/layers/attention_transducer.py:289-408,
    executed with its CUDA-only DelayTLoss bound to the reference's compiled CPU transducer: tests/golden/
    gen_golden_transducer_out.py) at delay_scale = 0: total / transducer / cross-entropy losses, d x and d W, for 1 and 3
    micro-batches and under a loss scaler.  Only the delay term itself stays unpinned."""
    z = np.load(os.path.join(GOLDEN, "transducer_out.npz"))
    out, dx, dW = R.transducer_out_step(z["x"], z["W"], z["targets"], z["src_len"], z["tgt_len"], delay_scale=0.0, temperature=1.0,
                                        label_smoothing=0.1, pad=1, ce_scale=1.0, delay_func="zero",
                                        loss_scale=float(z[f"{tag}.loss_scale"]), tokens_per_step=int(z[f"{tag}.tokens_per_step"]))
    B, T, U = z["x"].shape[:3]
    assert len(range(0, B, max(int(z[f"{tag}.tokens_per_step"]) // (T * U), 1))) == int(z[f"{tag}.micro_batches"])
    np.testing.assert_allclose(out["loss"], float(z[f"{tag}.loss"]), rtol=2e-6)
    np.testing.assert_allclose(out["loss_prob"], float(z[f"{tag}.loss_prob"]), rtol=2e-6)
    np.testing.assert_allclose(out["nll_loss"], float(z[f"{tag}.nll_loss"]), rtol=2e-6)
    assert int((z["targets"] != 1).sum()) == int(z[f"{tag}.sample_size"])
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))      # noqa: E731
    # measured 2.2e-6 (the reference runs fp32, the oracle fp64)
    assert rel(dx, z[f"{tag}.dx"]) < 1e-5 and rel(dW, z[f"{tag}.dW"]) < 1e-5, (rel(dx, z[f"{tag}.dx"]), rel(dW, z[f"{tag}.dW"]))
