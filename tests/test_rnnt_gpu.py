"""Row f4 on the GPU: the HIP transducer loss (csrc/rnnt.hip) through the reference's own C interface
(include/w2vs_rnnt.h = warp_transducer/include/rnnt.h) and through the PyTorch front end mirror, against
 (a) the known answers the reference's tests hold, (b) the reference's CPU implementation compiled into oracle/_ref,
 (c) the float64 oracle restatement of the CUDA kernels, (d) size-independent properties at a realistic size.
Needs an MI355X: pytest -m gpu.   Tolerance: 1e-4 (what the reference's tests use), fp32 arithmetic."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

import rnnt_oracle as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ka(golden_dir):
    return json.load(open(os.path.join(golden_dir, "rnnt_known_answers.json")))


def _dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def _c_api(acts, labels, xl, yl, blank=0, delay=None, delay_scale=1.0, smooth=1.0, want_grad=True):
    """Drive the reference-named entry points exactly as warp_transducer/tests/test_delay.cu does."""
    from wav2vec_s_amd import transducer as tr
    lib = tr._rnnt_lib()
    B, T, U, V = acts.shape
    a = _dev(acts, torch.float32)
    g = torch.full_like(a, float("nan")) if want_grad else None
    lab, x, y = _dev(labels, torch.int32), _dev(xl, torch.int32), _dev(yl, torch.int32)
    if lab.numel() == 0:                              # U = 1: no labels, but the interface (like the reference's) wants a pointer
        lab = torch.zeros(1, dtype=torch.int32, device="cuda")
    size = C.c_size_t(0)
    opt = tr.RnntOptions(1, 1, torch.cuda.current_stream().cuda_stream, blank, T, U, False)
    if delay is None:
        assert lib.get_workspace_size(T, U, B, True, C.byref(size), 4) == 0
        ws = torch.empty(size.value, dtype=torch.uint8, device="cuda")
        costs = np.zeros(B, dtype=np.float32)
        rc = lib.compute_rnnt_loss(a.data_ptr(), g.data_ptr() if want_grad else None, lab.data_ptr(), y.data_ptr(), x.data_ptr(),
                                   V, B, costs.ctypes.data, ws.data_ptr(), opt)
    else:
        assert lib.get_delay_workspace_size(T, U, B, True, C.byref(size), 4) == 0
        ws = torch.empty(size.value, dtype=torch.uint8, device="cuda")
        costs = np.zeros(3 * B, dtype=np.float32)
        dv = _dev(delay, torch.float32)
        rc = lib.compute_rnnt_delay_loss(a.data_ptr(), g.data_ptr() if want_grad else None, lab.data_ptr(), y.data_ptr(),
                                         x.data_ptr(), dv.data_ptr(), V, B, costs.ctypes.data, ws.data_ptr(), delay_scale,
                                         smooth, opt)
    assert rc == 0, lib.rnntGetStatusString(rc)
    return costs, (g.cpu().numpy() if want_grad else None)


def _case(c):
    acts = np.array(c["acts"], dtype=np.float32).reshape(c["B"], c["T"], c["U"], c["V"])
    return acts, np.array(c["labels"]), np.array(c["input_lengths"]), np.array(c["label_lengths"])


def test_reference_known_answers_through_the_c_api(ka):
    from wav2vec_s_amd import transducer as tr
    assert tr._rnnt_lib().get_warprnnt_version() == 1
    acts, lab, xl, yl = _case(ka["small"])                                   # test_gpu.cu small_test
    costs, _ = _c_api(acts, lab, xl, yl, want_grad=False)
    assert abs(costs[0] - ka["small"]["expected_score"]) < 1e-4
    dv = R.delay_cost("zero", 1, 2, 3, xl, yl)
    costs3, _ = _c_api(acts, lab, xl, yl, delay=dv, want_grad=False)         # test_delay.cu small_test: scores[0]
    assert abs(costs3[0] - ka["small"]["expected_score"]) < 1e-4
    c = ka["options"]                                                        # options_test: scores and gradients
    acts, lab, xl, yl = _case(c)
    costs, g = _c_api(acts, lab, xl, yl)
    np.testing.assert_allclose(costs, c["expected_scores"], atol=1e-4)
    np.testing.assert_allclose(g.reshape(-1), c["expected_grads_wrt_acts"], atol=1e-4)
    dv = R.delay_cost("zero", 2, 4, 3, xl, yl)
    costs3, g3 = _c_api(acts, lab, xl, yl, delay=dv, delay_scale=0.0, smooth=1.0)     # grad_check's setting
    np.testing.assert_allclose(costs3[:2], c["expected_scores"], atol=1e-4)
    np.testing.assert_allclose(costs3[4:], costs3[:2], atol=1e-6)
    np.testing.assert_allclose(g3.reshape(-1), c["expected_grads_wrt_acts"], atol=1e-4)


def _random_case(rng, B, T, U, V, scale=1.5):
    acts = (rng.randn(B, T, U, V) * scale).astype(np.float32)
    xl = rng.randint(max(2, T // 2), T + 1, size=B); xl[0] = T
    yl = rng.randint(1, U, size=B); yl[-1] = U - 1
    lab = rng.randint(1, V, size=(B, U - 1))
    return acts, lab, xl, yl


@pytest.mark.skipif(not R.RefCpuRnnt.available(), reason="oracle/_ref/libwarprnnt_cpu.so not built")
def test_plain_rnnt_equals_compiled_reference_cpu():
    ref = R.RefCpuRnnt()
    rng = np.random.RandomState(0)
    for B, T, U, V in ((3, 17, 6, 11), (2, 40, 13, 32), (4, 9, 9, 5), (2, 70, 80, 12), (1, 1, 1, 8), (2, 5, 1, 7)):
        acts, lab, xl, yl = _random_case(rng, B, T, U, V) if U > 1 else (
            (rng.randn(B, T, U, V)).astype(np.float32), np.zeros((B, 0), dtype=np.int64), np.full(B, T), np.zeros(B, dtype=np.int64))
        want_c, want_g = ref.loss_and_act_grads(acts, lab, xl, yl)
        got_c, got_g = _c_api(acts, lab, xl, yl)
        np.testing.assert_allclose(got_c, want_c, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(got_g, want_g, atol=1e-4)
        for b in range(B):                                                   # zeros outside T x U (the reference memsets)
            assert not got_g[b, xl[b]:].any() and not got_g[b, :, yl[b] + 1:].any()


@pytest.mark.parametrize("kind,smooth,scale", [("zero", 1.0, 1.0), ("diagonal", 1.0, 0.3), ("diag_positive", 0.7, 2.0),
                                               ("zero", 1.3, 0.0)])
def test_delay_transducer_equals_oracle(kind, smooth, scale):
    from wav2vec_s_amd import transducer as tr
    rng = np.random.RandomState(3)
    for B, T, U, V in ((3, 12, 5, 9), (2, 33, 17, 24), (2, 6, 70, 8)):
        acts, lab, xl, yl = _random_case(rng, B, T, U, V)
        dv = R.delay_cost(kind, B, T, U, xl, yl)
        want_c, want_g = R.delay_loss(acts, lab, xl, yl, dv, delay_scale=scale, smooth=smooth)
        got_c, got_g = _c_api(acts, lab, xl, yl, delay=dv, delay_scale=scale, smooth=smooth)
        # fp32 lattice values reach |alpha| ~ 200 for 70 labels (ulp 1.5e-5); the path weights exp(emit - alpha) inherit that
        # and the expected delay (~30) moves by ~1e-3 - an fp32 evaluation of the reference's own recursion on the CPU
        # lands on the same digits (33.1399 vs 33.1390 in float64 for this very case).  Short lattices: 2e-4.
        tol = 2e-4 if T + U <= 50 else 1.5e-3
        np.testing.assert_allclose(got_c.reshape(3, B), want_c, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(got_g, want_g, atol=tol * max(1.0, scale))
        # the front end: device costs, gradient in backward, both index conventions, upstream gradient folded in
        a = _dev(acts, torch.float32).requires_grad_(True)
        dvt = getattr(tr, "delay_cost_" + kind)(a, _dev(xl, torch.int32), _dev(yl, torch.int32))
        np.testing.assert_allclose(dvt.cpu().numpy(), dv, rtol=1e-6, atol=1e-6)
        for consistent in (False, True):
            loss = tr.DelayTLoss(blank=0, delay_scale=scale, temperature=smooth, reduction="none", delay_func=kind,
                                 consistent_delay_index=consistent)
            tot, nll, dly = loss(a, _dev(lab, torch.int32), _dev(xl, torch.int32), _dev(yl, torch.int32))
            assert tot.is_cuda and tot.shape == (B,)
            np.testing.assert_allclose(torch.stack([nll, dly, tot]).detach().cpu().numpy(), want_c, rtol=2e-4, atol=2e-4)
            up = torch.linspace(0.5, 2.0, B).cuda()
            a.grad = None
            (tot * up).sum().backward()
            _, wg = R.delay_loss(acts, lab, xl, yl, dv, delay_scale=scale, smooth=smooth, consistent_delay_index=consistent)
            np.testing.assert_allclose(a.grad.cpu().numpy(), wg * up.cpu().numpy()[:, None, None, None], atol=2 * tol * max(1.0, scale))


def test_front_end_reductions_checks_and_rnnt_loss():
    from wav2vec_s_amd import transducer as tr
    from wav2vec_s_amd._lib import W2vsError
    rng = np.random.RandomState(5)
    B, T, U, V = 3, 10, 4, 6
    acts, lab, xl, yl = _random_case(rng, B, T, U, V)
    xl[:] = [T, 7, 9]; yl[:] = [2, U - 1, 1]
    want_c, want_g = R.rnnt_loss(acts, lab, xl, yl)
    a = _dev(acts, torch.float32).requires_grad_(True)
    L, X, Y = _dev(lab, torch.int32), _dev(xl, torch.int32), _dev(yl, torch.int32)
    for red, cw, gw in (("mean", want_c.sum() / B, want_g / B), ("sum", want_c.sum(), want_g)):
        a.grad = None
        out = tr.RNNTLoss(blank=0, reduction=red)(a, L, X, Y)
        assert out.shape == (1,) and out.is_cuda
        np.testing.assert_allclose(out.item(), cw, rtol=1e-4)
        out.backward()
        np.testing.assert_allclose(a.grad.cpu().numpy(), gw, atol=1e-4)
    out = tr.rnnt_loss(a, L, X, Y, reduction="none")
    np.testing.assert_allclose(out.detach().cpu().numpy(), want_c, rtol=1e-4)
    tot, nll, dly = tr.DelayTLoss(reduction="mean")(a, L, X, Y)
    np.testing.assert_allclose(nll.item(), want_c.sum() / B, rtol=1e-4)
    with pytest.raises(TypeError, match="labels must be torch.int32"):
        tr.rnnt_loss(a, L.long(), X, Y)
    with pytest.raises(ValueError, match="Input length mismatch"):
        tr.rnnt_loss(a, L, X - 1, Y)
    with pytest.raises(ValueError, match="Output length mismatch"):
        tr.rnnt_loss(a, L, X, torch.ones_like(Y))
    with pytest.raises(ValueError, match="contiguous"):
        tr.rnnt_loss(a.transpose(1, 2), L, X, Y)
    with pytest.raises(NotImplementedError):
        tr.DelayTLoss()(a.detach().cpu(), L.cpu(), X.cpu(), Y.cpu())
    with pytest.raises(NotImplementedError):
        tr.DelayTLoss(delay_func="cubic")
    with pytest.raises(W2vsError):
        tr.rnnt_loss(a.detach().cpu(), L.cpu(), X.cpu(), Y.cpu())
    lib = tr._rnnt_lib()                                                     # the C interface rejects what the reference rejects
    opt = tr.RnntOptions(0, 1, None, 0, T, U, True)
    costs = np.zeros(B, dtype=np.float32)
    assert lib.compute_rnnt_loss(a.data_ptr(), None, L.data_ptr(), Y.data_ptr(), X.data_ptr(), V, B, costs.ctypes.data,
                                 a.data_ptr(), opt) == 3                    # RNNT_CPU: no CPU path -> EXECUTION_FAILED
    opt = tr.RnntOptions(1, 1, None, 0, T, U, True)
    assert lib.compute_rnnt_loss(None, None, L.data_ptr(), Y.data_ptr(), X.data_ptr(), V, B, costs.ctypes.data,
                                 a.data_ptr(), opt) == 2                    # INVALID_VALUE
    assert lib.rnntGetStatusString(2) == b"invalid value"


def test_speech_translation_sized_batch_properties():
    """B=8, T=160 frames, U=48 tokens, V=8000 (a CAAT joint output): no oracle at this size; check what must hold.
    Every valid row's gradient sums to zero (occupancy in = transition mass out), rows outside T x U are zero, the
    costs equal those of a V-reduced problem solved by the compiled reference, nothing is NaN/Inf."""
    from wav2vec_s_amd import transducer as tr
    g = torch.Generator().manual_seed(0)
    B, T, U, V = 8, 160, 48, 8000
    acts = (torch.randn(B, T, U, V, generator=g) * 2).cuda().requires_grad_(True)
    xl = torch.tensor([160, 151, 120, 99, 160, 80, 143, 160], dtype=torch.int32)
    yl = torch.tensor([47, 30, 12, 40, 1, 25, 47, 33], dtype=torch.int32)
    lab = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32)
    tot, nll, dly = tr.DelayTLoss(delay_scale=1.0, reduction="sum", delay_func="zero")(acts, lab.cuda(), xl.cuda(), yl.cuda())
    tot.backward()
    gr = acts.grad
    assert bool(torch.isfinite(gr).all()) and bool(torch.isfinite(tot))
    none = tr.rnnt_loss(acts.detach().requires_grad_(True), lab.cuda(), xl.cuda(), yl.cuda(), reduction="none")
    np.testing.assert_allclose(none.sum().item(), nll.item(), rtol=1e-5)
    a2 = acts.detach().clone().requires_grad_(True)
    tr.rnnt_loss(a2, lab.cuda(), xl.cuda(), yl.cuda(), reduction="sum").backward()
    rs = a2.grad.sum(-1)
    assert float(rs.abs().max()) < 2e-4
    for b in range(B):
        assert not bool(a2.grad[b, xl[b]:].any()) and not bool(a2.grad[b, :, yl[b] + 1:].any())
        assert not bool(gr[b, xl[b]:].any()) and not bool(gr[b, :, yl[b] + 1:].any())
    # expected delay: a sum of s / src_len over emitted labels
    dl = tr.DelayTLoss(delay_scale=1.0, reduction="none")(acts.detach(), lab.cuda(), xl.cuda(), yl.cuda())[2].cpu()
    assert bool((dl > 0).all()) and bool((dl < yl.float()).all())
    if R.RefCpuRnnt.available():                       # the loss only sees log p(blank) and log p(label): check it via the reference
        b = 2
        Tb, Ub = int(xl[b]), int(yl[b]) + 1
        lp = torch.log_softmax(acts.detach()[b:b + 1, :Tb, :Ub].double(), -1).cpu().numpy()
        want, _ = R.RefCpuRnnt().loss_and_logprob_grads(lp, lab[b:b + 1, :Ub - 1].numpy(), [Tb], [Ub - 1])
        np.testing.assert_allclose(none[b].item(), want[0], rtol=2e-4)


def _gen_labels(rng, V, L):
    """warp_transducer/tests/random.cpp genLabels: uniform labels in [1, V-1] with guaranteed repeats."""
    lab = rng.randint(1, V, size=L)
    if L >= 3:
        lab[L // 2] = lab[L // 2 + 1]
        lab[L // 2 - 1] = lab[L // 2]
    return lab


def test_reference_inf_test_and_grad_check_mirrors():
    """test_delay.cu inf_test (:247-333: V=15, T=50, L=10, activations uniform in [0,1): finite cost, no NaN) and
    run_tests / grad_check (:463-500: (V,T,L,B) = (20,50,15,1) and (5,10,5,65), analytic gradient against a central
    difference of the TOTAL cost with delay_scale = 0, epsilon 1e-2, rel_diff < 1e-2), through the same C entry point."""
    rng = np.random.RandomState(0)
    V, T, L = 15, 50, 10
    acts = rng.uniform(0, 1, size=(1, T, L, V)).astype(np.float32)
    lab = _gen_labels(rng, V, L - 1)[None]
    lab[0, 0] = 2
    dv = R.delay_cost("zero", 1, T, L, [T], [L - 1])
    costs, g = _c_api(acts, lab, [T], [L - 1], delay=dv, delay_scale=1.0, smooth=1.0)
    assert np.isfinite(costs).all() and not np.isnan(g).any()
    for V, T, L, B in ((20, 50, 15, 1), (5, 10, 5, 65)):
        acts = rng.uniform(0, 1, size=(B, T, L, V)).astype(np.float32)
        lab = np.stack([_gen_labels(rng, V, L - 1) for _ in range(B)])
        xl, yl = np.full(B, T), np.full(B, L - 1)
        dv = R.delay_cost("zero", B, T, L, xl, yl)
        _, g = _c_api(acts, lab, xl, yl, delay=dv, delay_scale=0.0, smooth=1.0)
        idx = rng.choice(acts.size, size=160, replace=False)     # a random subset of the coordinates grad_check sweeps
        eps = 1e-2
        num = np.zeros(len(idx))
        for k, i in enumerate(idx):
            ap, am = acts.copy().reshape(-1), acts.copy().reshape(-1)
            ap[i] += eps
            am[i] -= eps
            cp, _ = _c_api(ap.reshape(acts.shape), lab, xl, yl, delay=dv, delay_scale=0.0, want_grad=False)
            cm, _ = _c_api(am.reshape(acts.shape), lab, xl, yl, delay=dv, delay_scale=0.0, want_grad=False)
            num[k] = (cp[2 * B:].astype(np.float64).sum() - cm[2 * B:].astype(np.float64).sum()) / (2 * eps)
        ana = g.reshape(-1)[idx]
        rel_diff = ((ana - num) ** 2).sum() / (ana ** 2).sum()     # tests/test.h:22-32
        assert rel_diff < 1e-2, (V, T, L, B, rel_diff)


@pytest.mark.parametrize("tokens_per_step,scaled", [(20000, False), (60, True)])
def test_transducer_out_head_equals_oracle(tokens_per_step, scaled):
    """TransducerOut.train_step / eval_step (rain/layers/attention_transducer.py:289-446): projection + delay transducer +
    label-smoothed CE on the last frame, micro-batched (tokens_per_step = 60 forces 1 sample per micro-batch here), loss
    scaling, gradient of the projection weight and of the joint states pushed into the upstream graph."""
    from wav2vec_s_amd import transducer as tr
    rng = np.random.RandomState(11)
    B, T, U, d, V = 3, 9, 5, 16, 24
    x = torch.tensor(rng.randn(B, T, U, d), dtype=torch.float32).to(torch.bfloat16)
    W = torch.tensor(rng.randn(V, d) * 0.4, dtype=torch.float32).to(torch.bfloat16)
    tg = rng.randint(2, V, size=(B, U - 1)); tg[1, 3] = 1; tg[2, 2:] = 1
    sl, tl = np.array([9, 7, 5]), np.array([4, 3, 2])
    scale = 4.0 if scaled else 1.0
    want, wdx, wdW = R.transducer_out_step(x.float().numpy(), W.float().numpy(), tg, sl, tl, delay_scale=0.8, temperature=1.0,
                                           label_smoothing=0.1, pad=1, ce_scale=0.5, loss_scale=scale,
                                           tokens_per_step=tokens_per_step)
    proj = torch.nn.Linear(d, V, bias=False).to(torch.bfloat16).cuda()
    with torch.no_grad():
        proj.weight.copy_(W)
    head = tr.TransducerOut(proj, delay_scale=0.8, tokens_per_step=tokens_per_step, blank=0, label_smoothing=0.1, pad=1,
                            ce_scale=0.5, temperature=1.0)
    up = torch.nn.Parameter(x.clone().cuda())                # stands for the joint network: x = 1.0 * up
    xin = up * 1.0

    class _Scaler:
        def get_scale(self):
            return scale
    res = head.train_step(xin, torch.tensor(tg).cuda(), torch.tensor(sl).cuda(), torch.tensor(tl).cuda(),
                          scaler=_Scaler() if scaled else None)
    for k in ("loss", "loss_prob", "loss_delay", "nll_loss"):
        assert res[k].is_cuda
        np.testing.assert_allclose(float(res[k]), want[k], rtol=2e-3, atol=2e-3)
    assert res["sample_size"] == int((tg != 1).sum())
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12))      # noqa: E731
    assert rel(up.grad.float().cpu().numpy(), wdx) < 1e-2
    assert rel(proj.weight.grad.float().cpu().numpy(), wdW) < 1e-2
    ev = head.eval_step(up.detach(), torch.tensor(tg).cuda(), torch.tensor(sl).cuda(), torch.tensor(tl).cuda())
    np.testing.assert_allclose(float(ev["loss"]), want["loss"], rtol=2e-3, atol=2e-3)
    logits = head(up.detach())
    assert logits.shape == (B, T, U, V) and logits.dtype == torch.bfloat16
    assert rel(logits.float().cpu().numpy(), x.float().numpy() @ W.float().numpy().T) < 1e-2
    with pytest.raises(Exception, match="bias-free"):
        tr.TransducerOut(torch.nn.Linear(d, V, bias=True))


@pytest.mark.skipif(not R.RefCpuRnnt.available(), reason="oracle/_ref/libwarprnnt_cpu.so not built")
def test_nonzero_blank_label():
    """options.blank_label != 0 (rnnt.h:56): against the compiled reference CPU (plain) and the oracle (delay)."""
    ref = R.RefCpuRnnt()
    rng = np.random.RandomState(21)
    B, T, U, V = 3, 14, 6, 12
    acts = (rng.randn(B, T, U, V) * 1.3).astype(np.float32)
    xl, yl = np.array([14, 9, 12]), np.array([5, 3, 4])
    for blank in (V - 1, 4):
        lab = rng.randint(0, V - 1, size=(B, U - 1))
        lab[lab >= blank] += 1                                   # any label but the blank
        want_c, want_g = ref.loss_and_act_grads(acts, lab, xl, yl, blank=blank)
        got_c, got_g = _c_api(acts, lab, xl, yl, blank=blank)
        np.testing.assert_allclose(got_c, want_c, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(got_g, want_g, atol=1e-4)
        dv = R.delay_cost("diagonal", B, T, U, xl, yl)
        oc, og = R.delay_loss(acts, lab, xl, yl, dv, delay_scale=0.6, smooth=1.0, blank=blank)
        gc, gg = _c_api(acts, lab, xl, yl, blank=blank, delay=dv, delay_scale=0.6)
        np.testing.assert_allclose(gc.reshape(3, B), oc, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(gg, og, atol=2e-4)


def test_fp64_entry_equals_the_fp32_entry():
    """compute_rnnt_loss_fp64 (rnnt.h:115-124) is a converting wrapper: same costs and gradients as compute_rnnt_loss on the
    narrowed activations, to fp32 rounding - with gradients (the fp64 gradient buffer doubles as staging and is widened
    in place: an odd element count exercises every halving pass) and costs-only."""
    from wav2vec_s_amd import transducer as tr
    lib = tr._rnnt_lib()
    lib.compute_rnnt_loss_fp64.restype = C.c_int
    _fp64_case(3, 7, 5, 11)                                              # 1155 elements: odd -> allocated activation staging
    _fp64_case(3, 7, 5, 12)                                              # 1260 = 4 x 315: staged inside the gradient buffer, odd passes


def _fp64_case(B, T, U, V):
    from wav2vec_s_amd import transducer as tr
    lib = tr._rnnt_lib()
    rng = np.random.default_rng(5)
    acts = rng.standard_normal((B, T, U, V)).astype(np.float32)
    lab = rng.integers(1, V, size=(B, U - 1)).astype(np.int32)
    xl, yl = np.array([7, 5, 6], dtype=np.int32), np.array([4, 2, 3], dtype=np.int32)
    c32, g32 = _c_api(acts, lab, xl, yl)
    a64 = _dev(acts.astype(np.float64), torch.float64)
    g64 = torch.full_like(a64, float("nan"))
    labd, x, y = _dev(lab, torch.int32), _dev(xl, torch.int32), _dev(yl, torch.int32)
    size = C.c_size_t(0)
    assert lib.get_workspace_size(T, U, B, True, C.byref(size), 8) == 0
    ws = torch.empty(size.value, dtype=torch.uint8, device="cuda")
    opt = tr.RnntOptions(1, 1, torch.cuda.current_stream().cuda_stream, 0, T, U, False)
    for want_grad in (True, False):
        costs = np.zeros(B, dtype=np.float64)
        rc = lib.compute_rnnt_loss_fp64(C.c_void_p(a64.data_ptr()), C.c_void_p(g64.data_ptr()) if want_grad else None,
                                        C.c_void_p(labd.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), V, B,
                                        costs.ctypes.data_as(C.c_void_p), C.c_void_p(ws.data_ptr()), opt)
        assert rc == 0
        np.testing.assert_allclose(costs, c32, rtol=1e-6, atol=1e-6)
        if want_grad:
            np.testing.assert_allclose(g64.cpu().numpy(), g32, rtol=0, atol=1e-7)


@pytest.mark.parametrize("tag", ["mb1", "mb3", "mb3_scaled"])
def test_transducer_out_head_matches_reference_fixture(tag, golden_dir):
    """The HIP loss head against values recorded from the REFERENCE's TransducerOut.train_step at delay_scale = 0
    (tests/golden/transducer_out.npz <- gen_golden_transducer_out.py: rain/layers/attention_transducer.py:289-408 executed with
    DelayTLoss bound to the reference's compiled CPU transducer): losses, d x, d W for 1 and 3 micro-batches, label
    smoothing 0.1, and with the reference's scaler protocol (an object that only has .scale(loss))."""
    from wav2vec_s_amd import transducer as tr
    z = np.load(os.path.join(golden_dir, "transducer_out.npz"))
    B, T, U, d, V = [int(v) for v in z["cfg"]]
    proj = torch.nn.Linear(d, V, bias=False).to(torch.bfloat16).cuda()
    with torch.no_grad():
        proj.weight.copy_(torch.from_numpy(z["W"]))                       # bf16-representable values
    head = tr.TransducerOut(proj, delay_scale=0.0, tokens_per_step=int(z[f"{tag}.tokens_per_step"]), blank=0, label_smoothing=0.1,
                            delay_func="zero", pad=1, ce_scale=1.0, temperature=1.0)
    up = torch.nn.Parameter(torch.from_numpy(z["x"]).to(torch.bfloat16).cuda())
    scale = float(z[f"{tag}.loss_scale"])

    class _Scaler:                                                         # attention_transducer.py:397-398 calls scaler.scale(loss)
        def scale(self, loss):
            return loss * scale
    res = head.train_step(up * 1.0, torch.from_numpy(z["targets"]).cuda(), torch.from_numpy(z["src_len"]).cuda(),
                          torch.from_numpy(z["tgt_len"]).cuda(), scaler=_Scaler() if scale != 1.0 else None)
    for k in ("loss", "loss_prob", "nll_loss"):
        np.testing.assert_allclose(float(res[k]), float(z[f"{tag}.{k}"]), rtol=2e-3)
    assert res["sample_size"] == int(z[f"{tag}.sample_size"])
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12))      # noqa: E731
    assert rel(up.grad.float().cpu().numpy(), z[f"{tag}.dx"]) < 1e-2
    assert rel(proj.weight.grad.float().cpu().numpy(), z[f"{tag}.dW"]) < 1e-2
