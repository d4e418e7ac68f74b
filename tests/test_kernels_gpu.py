"""Per-kernel parity: every C-ABI entry of libw2vs against the CPU oracle / fp32 torch math on
the same seeded inputs.  Needs a real MI355X:  pytest -m gpu"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import w2vs_oracle as O

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    import wav2vec_s_amd  # noqa: F401
    from wav2vec_s_amd import ops as _ops
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return _ops


def dev(t):
    return t.cuda()


def rel(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 320, 136), (6544, 768, 768), (1000, 2304, 768), (257, 640, 512)])
def test_gemm_nt_bias(ops, M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3)
    y = ops.linear_fwd(dev(x), dev(w), dev(b))
    ref = x.float() @ w.float().t() + b.float()
    assert rel(y, ref) < 5e-3


def test_gemm_nt_gelu_save(ops):
    M, N, K = 300, 256, 192
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    y, pre = ops.linear_fwd(dev(x), dev(w), dev(b), gelu=True, save_pre=True)
    ref_pre = x.float() @ w.float().t() + b.float()
    assert rel(pre, ref_pre) < 5e-3
    assert rel(y, F.gelu(pre.float().cpu())) < 5e-3
    y2 = ops.linear_fwd(dev(x), dev(w), dev(b), gelu=True)
    assert rel(y2, F.gelu(ref_pre)) < 6e-3


def test_gemm_nt_dgrad_dgelu(ops):
    R, N, K = 260, 384, 256
    dy, w, pre = rnd(R, N, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(R, K, seed=3)
    wt = ops.transpose2d(dev(w))
    assert torch.equal(wt.cpu(), w.t().contiguous())
    dx = ops.linear_dgrad(dev(dy), wt)
    assert rel(dx, dy.float() @ w.float()) < 5e-3
    dx2 = ops.linear_dgrad(dev(dy), wt, dgelu_aux=dev(pre))
    p = pre.float().requires_grad_(True)
    F.gelu(p).backward(dx.float().cpu())  # gelu' applied to the bf16-rounded plain dgrad
    assert rel(dx2, p.grad) < 6e-3


@pytest.mark.parametrize("R,N,K", [(64, 128, 128), (1000, 256, 192), (6544, 768, 768), (333, 640, 512)])
def test_gemm_tn_wgrad(ops, R, N, K):
    dy, x = rnd(R, N, seed=1), rnd(R, K, seed=2)
    dw = torch.zeros(N, K, device="cuda")
    dbf = torch.zeros(N, device="cuda")
    ops.linear_wgrad(dev(dy), dev(x), dw, alpha=0.5, db_f32=dbf)
    ref = 0.5 * dy.float().t() @ x.float()
    assert rel(dw, ref) < 2e-3
    assert rel(dbf, 0.5 * dy.float().sum(0)) < 1e-3      # bias gradient fused into the wgrad GEMM
    db = torch.zeros(N, device="cuda")
    ops.colsum(dev(dy), db)
    assert rel(db, dy.float().sum(0)) < 1e-3


# ------------------------------------------------------------------------------------------ conv
@pytest.mark.parametrize("k,s,Lin", [(3, 2, 401), (3, 2, 400), (2, 2, 300), (2, 2, 301)])
def test_conv_channel_last(ops, k, s, Lin):
    B, Cin, Cout = 3, 64, 128
    x = rnd(B, Lin, Cin, seed=1)
    w = rnd(Cout, Cin, k, seed=2, scale=0.1)
    w2 = ops.conv_pack_weight(dev(w))
    assert torch.equal(w2.cpu(), w.permute(0, 2, 1).reshape(Cout, k * Cin))
    y, pre = ops.conv_cl_fwd(dev(x), w2, k, s)
    xr = x.float().transpose(1, 2).requires_grad_(True)
    wr = w.float().requires_grad_(True)
    ref_pre = F.conv1d(xr, wr, stride=s)
    assert rel(pre, ref_pre.transpose(1, 2)) < 5e-3
    assert rel(y, F.gelu(pre.float().cpu())) < 5e-3
    Lout = ref_pre.shape[-1]
    dy = rnd(B, Lout, Cout, seed=3)
    ref_pre.backward(dy.float().transpose(1, 2))
    dx = ops.conv_cl_dgrad(dev(dy), w2, k, s, Lin)
    assert rel(dx, xr.grad.transpose(1, 2)) < 5e-3
    dw2 = torch.zeros(Cout, k * Cin, device="cuda")
    ops.conv_cl_wgrad(dev(dy), dev(x), k, s, dw2)
    assert rel(dw2, wr.grad.permute(0, 2, 1).reshape(Cout, k * Cin)) < 3e-3
    # dgrad chained through the previous layer's GELU
    aux = rnd(B, Lin, Cin, seed=4)
    dx2 = ops.conv_cl_dgrad(dev(dy), w2, k, s, Lin, dgelu_aux=dev(aux))
    a = aux.float().requires_grad_(True)
    F.gelu(a).backward(dx.float().cpu())
    assert rel(dx2, a.grad) < 6e-3


def test_conv0_ln_gelu(ops):
    B, L, Cc, k, s = 2, 4003, 512, 10, 5
    wave = rnd(B, L, seed=1)
    w = rnd(Cc, 1, k, seed=2, scale=0.4)
    g, b = (1 + 0.1 * torch.randn(Cc)).to(BF), (0.1 * torch.randn(Cc)).to(BF)
    y, mean, rstd = ops.conv0_fwd(dev(wave), dev(w), dev(g), dev(b), k, s)
    P = {"feature_extractor.conv_layers.0.0.weight": w.float().requires_grad_(True),
         "feature_extractor.conv_layers.0.2.1.weight": g.float().requires_grad_(True),
         "feature_extractor.conv_layers.0.2.1.bias": b.float().requires_grad_(True)}
    cfg = O.OracleCfg(conv_feature_layers="[(512, 10, 5)]")
    ref = O.conv_feature_extractor(wave.float(), P, cfg)  # B x C x L0
    assert rel(y, ref.transpose(1, 2)) < 5e-3
    dy = rnd(*y.shape, seed=5)
    ref.backward(dy.float().transpose(1, 2))
    dw = torch.zeros(Cc, k, device="cuda")
    dg = torch.zeros(Cc, device="cuda")
    db = torch.zeros(Cc, device="cuda")
    ops.conv0_bwd(dev(wave), dev(w), dev(g), dev(b), mean, rstd, dev(dy), k, s, dw, dg, db)
    assert rel(dw, P["feature_extractor.conv_layers.0.0.weight"].grad.view(Cc, k)) < 5e-3
    assert rel(dg, P["feature_extractor.conv_layers.0.2.1.weight"].grad) < 5e-3
    assert rel(db, P["feature_extractor.conv_layers.0.2.1.bias"].grad) < 5e-3


def test_conv0_groupnorm_gelu(ops):
    """extractor_mode='default': conv layer 0 + Fp32GroupNorm(C, C) + GELU (wav2vec2.py:744-750)."""
    B, L, Cc, k, s = 3, 3003, 512, 10, 5
    wave = rnd(B, L, seed=1)
    w = rnd(Cc, 1, k, seed=2, scale=0.4)
    g, b = (1 + 0.1 * torch.randn(Cc)).to(BF), (0.1 * torch.randn(Cc)).to(BF)
    y, stat = ops.conv0_gn_fwd(dev(wave), dev(w), dev(g), dev(b), k, s)
    P = {"feature_extractor.conv_layers.0.0.weight": w.float().requires_grad_(True),
         "feature_extractor.conv_layers.0.2.weight": g.float().requires_grad_(True),
         "feature_extractor.conv_layers.0.2.bias": b.float().requires_grad_(True)}
    cfg = O.OracleCfg(conv_feature_layers="[(512, 10, 5)]", extractor_mode="default")
    ref = O.conv_feature_extractor(wave.float(), P, cfg)
    assert rel(y, ref.transpose(1, 2)) < 5e-3
    dy = rnd(*y.shape, seed=5)
    ref.backward(dy.float().transpose(1, 2))
    dw = torch.zeros(Cc, k, device="cuda"); dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    ops.conv0_gn_bwd(dev(wave), dev(w), dev(g), dev(b), stat, dev(dy), k, s, dw, dg, db)
    assert rel(dw, P["feature_extractor.conv_layers.0.0.weight"].grad.view(Cc, k)) < 8e-3
    assert rel(dg, P["feature_extractor.conv_layers.0.2.weight"].grad) < 5e-3
    assert rel(db, P["feature_extractor.conv_layers.0.2.bias"].grad) < 5e-3


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("Cc", [512, 768, 1024])
def test_layernorm_fwd_bwd(ops, Cc):
    rows = 777
    x, r = rnd(rows, Cc, seed=1), rnd(rows, Cc, seed=2)
    g, b = (1 + 0.1 * torch.randn(Cc)).to(BF), (0.1 * torch.randn(Cc)).to(BF)
    sumsq = torch.zeros(1, device="cuda")
    y, s_out, mean, rstd = ops.ln_fwd(dev(x), dev(g), dev(b), res=dev(r), want_sum=True, sumsq=sumsq)
    xs = (x.float() + r.float())
    assert rel(s_out, xs) < 4e-3
    sr = s_out.float().cpu().requires_grad_(True)
    gr, br = g.float().requires_grad_(True), b.float().requires_grad_(True)
    ref = F.layer_norm(sr, (Cc,), gr, br, 1e-5)
    assert rel(y, ref) < 4e-3
    assert abs(float(sumsq) - float((x.float() ** 2).sum())) / float((x.float() ** 2).sum()) < 1e-4
    dy = rnd(rows, Cc, seed=3)
    ref.backward(dy.float())
    dg = torch.zeros(Cc, device="cuda")
    db = torch.zeros(Cc, device="cuda")
    dx, dres = ops.ln_bwd(s_out, dev(g), dev(b), mean, rstd, dg, db, dy=dev(dy), want_dres=True)
    assert rel(dx, sr.grad) < 5e-3 and rel(dres, sr.grad) < 5e-3
    assert rel(dg, gr.grad) < 5e-3 and rel(db, br.grad) < 5e-3


@pytest.mark.parametrize("Cc,use_dsum,p_drop", [(768, False, 0.1), (768, True, 0.0), (1024, True, 0.1), (512, False, 0.0), (1024, False, 0.1)])
def test_layernorm_bwd_lean_one_row_per_wave(ops, Cc, use_dsum, p_drop):
    """The encoder layers' backward at activation sizes (rows >= 1024: ln_bwd_lean_kernel, one row per wave, 8-byte pieces,
    row sums folded through LDS into the partial slab): dx with the dropout decisions of the row kernels' hash applied, dres
    without them, the pre-LN form's extra stream gradient (dsum), dgamma / dbeta - against autograd of F.layer_norm.  2 501
    rows: the last workgroup is ragged (fs/modules/layer_norm.py:30-35, wav2vec2.py:955-976)."""
    rows = 2501
    x, r = rnd(rows, Cc, seed=1), rnd(rows, Cc, seed=2)
    g, b = (1 + 0.1 * torch.randn(Cc)).to(BF), (0.1 * torch.randn(Cc)).to(BF)
    y, s_out, mean, rstd = ops.ln_fwd(dev(x), dev(g), dev(b), res=dev(r), want_sum=True)
    sr = s_out.float().cpu().requires_grad_(True)
    gr, br = g.float().requires_grad_(True), b.float().requires_grad_(True)
    ref = F.layer_norm(sr, (Cc,), gr, br, 1e-5)
    assert rel(y, ref) < 4e-3
    dy = rnd(rows, Cc, seed=3)
    ds = rnd(rows, Cc, seed=4) if use_dsum else None
    ref.backward(dy.float())
    want_res = sr.grad + (ds.float() if use_dsum else 0.0)
    dg = torch.zeros(Cc, device="cuda")
    db = torch.zeros(Cc, device="cuda")
    dx, dres = ops.ln_bwd(s_out, dev(g), dev(b), mean, rstd, dg, db, dy=dev(dy), dsum=dev(ds) if use_dsum else None,
                          want_dres=True, p_drop=p_drop, seed=77)
    assert rel(dres, want_res) < 5e-3
    if p_drop > 0:
        keep = (ops.dropout(torch.ones(rows, Cc, device="cuda", dtype=BF), p_drop, 77).float() > 0).float().cpu()
        thr16 = int(float(np.float32(p_drop)) * 4294967296.0) >> 16
        want_dx = want_res * keep * (65536.0 / (65536.0 - thr16))
        assert 0.85 < float(keep.mean()) < 0.95
    else:
        want_dx = want_res
    assert rel(dx, want_dx) < 5e-3
    assert rel(dg, gr.grad) < 5e-3 and rel(db, br.grad) < 5e-3


def test_layernorm_gelu_penalty_scale(ops):
    """feat-LN backward: GradMultiply scale, features_pen term and the producer's GELU chained."""
    rows, Cc = 500, 512
    pre = rnd(rows, Cc, seed=1)
    g, b = (1 + 0.1 * torch.randn(Cc)).to(BF), (0.1 * torch.randn(Cc)).to(BF)
    p = pre.float().requires_grad_(True)
    x = F.gelu(p)
    xb = x.detach().to(BF)
    y, _, mean, rstd = ops.ln_fwd(dev(xb), dev(g), dev(b))
    xr = xb.float().requires_grad_(True)
    ref = F.layer_norm(xr, (Cc,), g.float(), b.float(), 1e-5)
    assert rel(y, ref) < 4e-3
    dy = rnd(rows, Cc, seed=2)
    pen_coef, scale = 0.37, 0.1
    (ref * dy.float()).sum().backward()
    want_dx = (xr.grad + 2 * xb.float() * pen_coef) * scale
    F.gelu(p).backward(want_dx)
    dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    dx, _ = ops.ln_bwd(dev(xb), dev(g), dev(b), mean, rstd, dg, db, dy=dev(dy), aux=dev(pre), out_scale=scale,
                       pen_coef=pen_coef)
    assert rel(dx, p.grad) < 8e-3


def test_dropout_statistics_and_consistency(ops):
    rows, Cc, p = 2048, 768, 0.1
    x = torch.ones(rows, Cc).to(BF)
    zero = torch.zeros(rows, Cc).to(BF)
    g, b = torch.ones(Cc).to(BF), torch.zeros(Cc).to(BF)
    _, s1, _, _ = ops.ln_fwd(dev(x), dev(g), dev(b), res=dev(zero), want_y=False, want_sum=True, p_drop=p, seed=1234)
    _, s2, _, _ = ops.ln_fwd(dev(x), dev(g), dev(b), res=dev(zero), want_y=False, want_sum=True, p_drop=p, seed=1234)
    _, s3, _, _ = ops.ln_fwd(dev(x), dev(g), dev(b), res=dev(zero), want_y=False, want_sum=True, p_drop=p, seed=99)
    assert torch.equal(s1, s2) and not torch.equal(s1, s3)
    keep = (s1.float() > 0).float().mean().item()
    assert abs(keep - 0.9) < 3e-3
    vals = torch.unique(s1.float().cpu())
    assert vals.tolist() == [0.0, float(torch.tensor(1 / 0.9).to(BF))]
    # backward uses the same mask: d(x) = mask/(1-p) * dy with LN bypassed (dsum path)
    y, s, mean, rstd = ops.ln_fwd(dev(x), dev(g), dev(b), res=dev(zero), want_sum=True, p_drop=p, seed=1234)
    dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    dx, _ = ops.ln_bwd(s, dev(g), dev(b), mean, rstd, dg, db, dy=None, dsum=dev(x), p_drop=p, seed=1234)
    assert torch.equal(dx, s1)


# ------------------------------------------------------------------------------------------ prologue
def _structure(Tp, m, r):
    rc_idx, rc_oob, masked = O.block_structure(Tp, m, r)
    src = torch.cat([torch.arange(Tp), rc_idx]).int()
    copies = [[] for _ in range(Tp)]
    for c, t in enumerate(rc_idx.tolist()):
        copies[t].append(Tp + c)
    start = np.zeros(Tp + 1, dtype=np.int32)
    for t in range(Tp):
        start[t + 1] = start[t] + len(copies[t])
    lst = np.array([c for cs in copies for c in cs] or [0], dtype=np.int32)
    return rc_idx, rc_oob, masked, src, torch.from_numpy(start), torch.from_numpy(lst)


@pytest.mark.parametrize("T,m,r,apply_ln", [(49, 8, 4, True), (50, 16, 8, True), (33, 8, 4, False)])
def test_enc_prologue(ops, T, m, r, apply_ln):
    B, Cc = 2, 768
    Tp = T + (T % 2)
    x = rnd(B, T, Cc, seed=1)
    mask = torch.zeros(B, T, dtype=torch.bool)
    mask[0, 3:13] = True
    mask[1, 20:30] = True
    mask_emb = rnd(Cc, seed=2)
    g, b = (1 + 0.1 * torch.randn(Cc)).to(BF), (0.1 * torch.randn(Cc)).to(BF)
    table = O.sinusoidal_table(8002, Cc, 1)
    pad = torch.zeros(B, T, dtype=torch.bool)
    pos = O.positions_from_padding(pad).int()
    rc_idx, rc_oob, masked, src, cstart, clist = _structure(Tp, m, r)
    out, mean, rstd = ops.enc_prologue_fwd(dev(x), dev(mask.to(torch.uint8)), None, dev(pos), dev(mask_emb), dev(table),
                                           dev(g), dev(b), dev(src), Tp, apply_ln=apply_ln)
    # oracle steps (wav2vec2.py:446, wav2vec_S.py:357-388, 465-484)
    xr = x.float().requires_grad_(True)
    me = mask_emb.float().requires_grad_(True)
    gr, br = g.float().requires_grad_(True), b.float().requires_grad_(True)
    v = torch.where(mask.unsqueeze(-1), me.view(1, 1, Cc).expand(B, T, Cc), xr)
    v = v + table.index_select(0, pos.long().view(-1)).view(B, T, Cc)
    if apply_ln:
        v = F.layer_norm(v, (Cc,), gr, br, 1e-5)
    v = F.pad(v, (0, 0, 0, Tp - T))
    full = torch.cat([v, v.index_select(1, rc_idx)], dim=1)
    assert rel(out, full) < 5e-3
    dout = rnd(*full.shape, seed=5)
    full.backward(dout.float())
    dme = torch.zeros(Cc, device="cuda"); dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    dx = ops.enc_prologue_bwd(dev(dout), dev(x), dev(mask.to(torch.uint8)), None, dev(pos), dev(mask_emb), dev(table),
                              dev(g), dev(b), mean, rstd, dev(src), dev(cstart), dev(clist), Tp, dme, dg, db,
                              apply_ln=apply_ln)
    assert rel(dx, xr.grad) < 6e-3
    assert rel(dme, me.grad) < 6e-3
    if apply_ln:
        assert rel(dg, gr.grad) < 6e-3 and rel(db, br.grad) < 6e-3


# ------------------------------------------------------------------------------------------ attention
def _dense_attention(qkv, H, Tp, m, r, kpad):
    """Reference semantics: additive -1e4 block mask + -inf key padding, softmax, P.V (oracle encoder_layer core)."""
    B, N, C3 = qkv.shape
    Cc = C3 // 3
    D = Cc // H
    q, k, v = qkv.float().split(Cc, dim=-1)
    q = q.view(B, N, H, D).permute(0, 2, 1, 3) * (D ** -0.5)
    k = k.view(B, N, H, D).permute(0, 2, 1, 3)
    v = v.view(B, N, H, D).permute(0, 2, 1, 3)
    _, _, masked = O.block_structure(Tp, m, r)
    add = torch.zeros(N, N).masked_fill(masked, -1e4).view(1, 1, N, N)
    if kpad is not None:
        add = add + torch.zeros(B, 1, 1, N).masked_fill(kpad.bool().view(B, 1, 1, N), float("-inf"))
    p = torch.softmax(q @ k.transpose(-1, -2) + add, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, N, Cc)


@pytest.mark.parametrize("Tp,m,r,H", [(48, 16, 8, 2), (50, 8, 4, 3), (130, 32, 16, 2), (546, 16, 8, 2), (40, 8, 0, 1),
                                      (10, 16, 8, 1), (300, 24, 6, 2), (97, 7, 3, 1), (200, 13, 5, 2), (64, 1, 1, 1)])
def test_block_attention(ops, Tp, m, r, H):
    B = 2
    N = Tp + (Tp // m) * r
    Cc = H * 64
    qkv = rnd(B, N, 3 * Cc, seed=Tp)
    rc_idx, rc_oob, _ = O.block_structure(Tp, m, r)
    pad = torch.zeros(B, Tp, dtype=torch.bool)
    pad[1, Tp - 1] = True
    kpad = pad
    if r > 0:
        kpad = torch.cat([pad, pad.index_select(1, rc_idx) | rc_oob.unsqueeze(0)], dim=1)
    o, lse = ops.attn_fwd(dev(qkv), H, Tp, m, r, kpad=dev(kpad.to(torch.uint8)))
    qr = qkv.float().requires_grad_(True)
    ref = _dense_attention(qr, H, Tp, m, r, kpad)
    assert rel(o, ref) < 8e-3
    dout = rnd(B, N, Cc, seed=7)
    ref.backward(dout.float())
    dqkv = ops.attn_bwd(dev(dout), dev(qkv), o, lse, H, Tp, m, r, kpad=dev(kpad.to(torch.uint8)))
    Cq = Cc
    for name, sl in [("dq", slice(0, Cq)), ("dk", slice(Cq, 2 * Cq)), ("dv", slice(2 * Cq, 3 * Cq))]:
        e = rel(dqkv[..., sl], qr.grad[..., sl])
        assert e < 1.5e-2, (name, e)


@pytest.mark.parametrize("Tp,m,r", [(999, 16, 8), (3000, 16, 8)])
def test_block_attention_long_lists_use_four_waves(ops, Tp, m, r):
    """attention2.hip runs forward / dQ with 4 waves per workgroup once the longest sub-tile list exceeds 32 (N = 1495:
    47) and the dK/dV pass with 4 once a key tile has more than 128 query sub-tiles (N = 4496: 141); the small shapes above
    all take the 2-wave instantiations.  Forward and backward against dense fp32 autograd on the GPU."""
    H, B = 1, 1
    N = Tp + (Tp // m) * r
    qkv = dev(rnd(B, N, 3 * 64, seed=Tp))
    o, lse = ops.attn_fwd(qkv, H, Tp, m, r)
    _, _, masked = O.block_structure(Tp, m, r)
    add = torch.zeros(N, N, device="cuda").masked_fill(masked.cuda(), -1e4)
    x = qkv[0].float().requires_grad_(True)
    q, k, v = x.split(64, dim=-1)
    ref = torch.softmax((q * 0.125) @ k.t() + add, dim=-1) @ v
    assert rel(o[0], ref) < 8e-3
    dout = dev(rnd(B, N, 64, seed=3))
    ref.backward(dout[0].float())
    dqkv = ops.attn_bwd(dout, qkv, o, lse, H, Tp, m, r)
    for name, sl in [("dq", slice(0, 64)), ("dk", slice(64, 128)), ("dv", slice(128, 192))]:
        e = rel(dqkv[0][..., sl], x.grad[..., sl])
        assert e < 1.5e-2, (name, e)


def test_block_attention_beyond_8192_positions_takes_the_first_kernels(ops):
    """attention2.hip's record table holds 256 tiles (N <= 8192); longer sequences run attention.hip's kernels behind the same
    entry.  Forward only against the dense reference (the 8244^2 score matrix is built once, on the GPU in fp32)."""
    Tp, m, r, H, B = 5500, 16, 8, 1, 1
    N = Tp + (Tp // m) * r
    assert N > 8192
    qkv = rnd(B, N, 3 * H * 64, seed=11)
    o, lse = ops.attn_fwd(dev(qkv), H, Tp, m, r)
    _, _, masked = O.block_structure(Tp, m, r)
    q, k, v = dev(qkv).float().split(64 * H, dim=-1)
    s = (q[0] * 0.125) @ k[0].t() + torch.zeros(N, N, device="cuda").masked_fill(masked.cuda(), -1e4)
    ref = torch.softmax(s, dim=-1) @ v[0]
    assert rel(o[0], ref) < 8e-3
    assert torch.isfinite(lse).all()


@pytest.mark.parametrize("Tp,m,r,H,nq", [(130, 32, 16, 2, 0), (546, 16, 8, 3, 0), (300, 24, 6, 2, 300)])
def test_attention_stored_keep_masks_equal_rehash(ops, Tp, m, r, H, nq):
    """w2vs_attn_desc.drop_bits: the forward parks its dropout decisions as bits, the backward reads them instead of
    re-hashing - same decisions, so every gradient must equal the recompute path's bit for bit (the formulas are shared)."""
    B = 2
    N = Tp + (Tp // m) * r
    qkv = dev(rnd(B, N, 3 * H * 64, seed=Tp))
    dout = dev(rnd(B, N, H * 64, seed=9))
    o1, lse1 = ops.attn_fwd(qkv, H, Tp, m, r, p_drop=0.2, seed=77)
    g1 = ops.attn_bwd(dout, qkv, o1, lse1, H, Tp, m, r, p_drop=0.2, seed=77)
    bits = ops.attn_drop_bits(B, H, N)
    bits.fill_(0x5A5A5A5A)                     # stale garbage must not matter: every block that is read was written
    o2, lse2 = ops.attn_fwd(qkv, H, Tp, m, r, p_drop=0.2, seed=77, drop_bits=bits)
    g2 = ops.attn_bwd(dout, qkv, o2, lse2, H, Tp, m, r, p_drop=0.2, seed=77, drop_bits=bits)
    assert torch.equal(o1, o2) and torch.equal(lse1, lse2)
    assert torch.equal(g1, g2)
    g3 = ops.attn_bwd(dout, qkv, o2, lse2, H, Tp, m, r, p_drop=0.2, seed=78, drop_bits=bits)   # the seed is not consulted
    assert torch.equal(g2, g3)


def test_attention_keep_decisions_equal_the_numpy_mirror(ops):
    """The decisions the forward kernel parks in drop_bits (= the ones every attention kernel hashes) against
    tests/hash_mirror.py, the numpy restatement of attn_common.h::pair_hash_pm whose statistics - within a seed and across
    seeds - tests/test_host_cpu.py checks: every VISIBLE (query, key) pair of three shapes and two 64-bit seeds."""
    import hash_mirror as Hm
    for (B, H, Tp, m, r, p, seed) in ((2, 3, 130, 32, 16, 0.1, 0x1234567890ABCDEF), (1, 2, 546, 16, 8, 0.25, 77),
                                      (2, 2, 96, 24, 6, 0.1, (0xFFFFFFFF << 32) | 5)):
        N = Tp + ((Tp + m - 1) // m - 1) * r if r > 0 else Tp
        from wav2vec_s_amd import host_rng
        N = host_rng.block_layout(Tp, m, r).N
        qkv = torch.zeros(B, N, 3 * H * 64, device="cuda", dtype=BF)
        bits = ops.attn_drop_bits(B, H, N)
        bits.fill_(-1)
        ops.attn_fwd(qkv, H, Tp, m, r, p_drop=p, seed=seed, drop_bits=bits)
        got = Hm.decode_drop_bits(bits, B, H, N).cpu().numpy()
        want = Hm.attn_keep(seed, B, H, N, p)
        _, _, masked = O.block_structure(Tp, m, r)
        vis = ~masked.numpy()                                   # [N, N] query x key
        assert vis.shape == (N, N)
        assert np.array_equal(got[:, :, vis], want[:, :, vis]), (B, H, Tp, m, r)
        rate = want[:, :, vis].mean()
        assert abs(rate - (1 - Hm.thr16(p) / 65536.0)) < 0.01


def test_attention_dropout_consistency(ops):
    """fwd and the two bwd passes regenerate the same keep-mask: finite-difference style check
    through linearity in V (O is linear in V for a fixed mask) and dV == P_drop^T dO."""
    B, H, Tp, m, r = 1, 1, 64, 16, 8
    N = Tp + (Tp // m) * r
    qkv = rnd(B, N, 192, seed=3)
    p = 0.25
    o1, lse = ops.attn_fwd(dev(qkv), H, Tp, m, r, p_drop=p, seed=42)
    o2, _ = ops.attn_fwd(dev(qkv), H, Tp, m, r, p_drop=p, seed=42)
    o3, _ = ops.attn_fwd(dev(qkv), H, Tp, m, r, p_drop=p, seed=43)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    # recover the dropped probability matrix 64 key columns at a time with one-hot V rows
    Pd = torch.zeros(N, N)
    for c0 in range(0, N, 64):
        w = min(64, N - c0)
        vv = qkv.float().clone()
        vv[..., 128:] = 0
        vv[0, c0:c0 + w, 128:128 + w] = torch.eye(w)
        oc, _ = ops.attn_fwd(dev(vv.to(BF)), H, Tp, m, r, p_drop=p, seed=42)
        Pd[:, c0:c0 + w] = oc[0].float().cpu()[:, :w]
    _, _, masked = O.block_structure(Tp, m, r)
    assert float(Pd[masked].abs().max()) == 0.0
    allowed = ~masked
    frac_zero = float((Pd[allowed] == 0).float().mean())
    assert abs(frac_zero - p) < 0.03
    dout = rnd(B, N, 64, seed=9)
    dqkv = ops.attn_bwd(dev(dout), dev(qkv), o1, lse, H, Tp, m, r, p_drop=p, seed=42)
    dv_ref = Pd.t() @ dout[0].float()
    assert rel(dqkv[0, :, 128:], dv_ref) < 2e-2


# ------------------------------------------------------------------------------------------ quantizer
@pytest.mark.parametrize("training", [True, False])
def test_gumbel_quantizer(ops, training):
    B, M, Fd, G, V, D = 2, 37, 512, 2, 320, 128
    cfg = O.OracleCfg()
    P = {"quantizer.weight_proj.weight": rnd(G * V, Fd, seed=1, scale=0.2).float(),
         "quantizer.weight_proj.bias": rnd(G * V, seed=2, scale=0.1).float(),
         "quantizer.vars": torch.rand(1, G * V, D).to(BF).float()}
    y = rnd(B, M, Fd, seed=3)
    logits = ops.linear_fwd(dev(y.view(-1, Fd)), dev(P["quantizer.weight_proj.weight"].to(BF)),
                            dev(P["quantizer.weight_proj.bias"].to(BF)))
    noise = None
    if training:
        g = torch.Generator().manual_seed(5)
        noise = -torch.empty(B * M * G, V).exponential_(generator=g).log()
    q, st = ops.quant_fwd(logits, dev(P["quantizer.vars"][0].to(BF)), G, V, 1.7, training,
                          noise=dev(noise) if training else None)
    # oracle on the SAME bf16 logits (the linear itself is checked by the GEMM tests)
    lg = logits.float().cpu().requires_grad_(True)
    vars_ = P["quantizer.vars"].clone().requires_grad_(True)
    Pq = dict(P)
    Pq["quantizer.vars"] = vars_

    def from_logits(lg):
        # gumbel_quantize with the projection replaced by identity on precomputed logits
        Pi = dict(Pq)
        Pi["quantizer.weight_proj.weight"] = torch.eye(G * V)
        Pi["quantizer.weight_proj.bias"] = torch.zeros(G * V)
        return O.gumbel_quantize(lg.view(B, M, G * V), Pi, cfg, 1.7, noise)

    qr, idx, prob_ppl, code_ppl = from_logits(lg)
    assert torch.equal(st.idx.cpu().long(), idx)
    assert rel(q, qr.view(B * M, G * D)) < 1e-6
    assert abs(float(st.ppl[0]) - float(prob_ppl)) / float(prob_ppl) < 1e-4
    assert abs(float(st.ppl[1]) - float(code_ppl)) / float(code_ppl) < 1e-4
    dq = rnd(B * M, G * D, seed=8)
    ppl_grad = -0.3
    ((qr.view(B * M, G * D) * dq.float()).sum() + ppl_grad * prob_ppl).backward()
    dvars = torch.zeros(G * V, D, device="cuda")
    dlogits = ops.quant_bwd(dev(dq), logits, dev(P["quantizer.vars"][0].to(BF)), st, G, V, 1.7, training, ppl_grad,
                            dvars, noise=dev(noise) if training else None)
    assert rel(dvars, vars_.grad[0]) < 2e-3
    assert rel(dlogits, lg.grad) < 1.5e-2


def test_gumbel_device_rng_statistics(ops):
    R, G, V, D = 4096, 2, 320, 128
    logits = torch.zeros(R, G * V).to(BF)
    vars2 = torch.rand(G * V, D).to(BF)
    q, st = ops.quant_fwd(dev(logits), dev(vars2), G, V, 2.0, True, seed=7)
    q2, st2 = ops.quant_fwd(dev(logits), dev(vars2), G, V, 2.0, True, seed=7)
    assert torch.equal(st.idx, st2.idx)
    cnt = torch.bincount(st.idx[:, 0].cpu().long(), minlength=V).float()
    # uniform logits + iid gumbel noise -> uniform argmax: chi-square well inside a loose bound
    chi2 = float(((cnt - R / V) ** 2 / (R / V)).sum())
    assert chi2 < 2 * V


# ------------------------------------------------------------------------------------------ InfoNCE
@pytest.mark.parametrize("B,M,K,Cc", [(2, 20, 10, 256), (3, 37, 100, 256), (2, 16, 7, 768)])
def test_infonce_logits(ops, B, M, K, Cc):
    cfg = O.OracleCfg()
    x, y = rnd(B, M, Cc, seed=1), rnd(B, M, Cc, seed=2)
    y[0, 3] = y[0, 5]  # a negative equal to the positive -> -inf (wav2vec2.py:531, 539-540)
    torch.manual_seed(0)
    neg = O.sample_negative_indices(B, M, K)
    neg[0, 3 * K] = 5
    logits, norms = ops.nce_fwd(dev(x.view(-1, Cc)), dev(y.view(-1, Cc)), dev(neg), B, M, K, cfg.logit_temp)
    xr, yr = x.float().requires_grad_(True), y.float().requires_grad_(True)
    preds, ref = O.compute_logits(xr, yr, neg, cfg)
    ref_bm = preds.permute(1, 2, 0).reshape(B * M, K + 1)  # rows (b, m)
    inf_mask = torch.isinf(ref_bm)
    assert torch.equal(torch.isinf(logits.cpu()), inf_mask) and bool(inf_mask.any())
    assert float((logits.cpu()[~inf_mask] - ref_bm[~inf_mask]).abs().max()) < 2e-3
    dl = torch.randn(B * M, K + 1)
    dl[inf_mask] = 0
    (ref_bm.masked_fill(inf_mask, 0) * dl).sum().backward()
    dx, dy = ops.nce_bwd(dev(dl), logits, norms, dev(x.view(-1, Cc)), dev(y.view(-1, Cc)), dev(neg), B, M, K, cfg.logit_temp)
    assert rel(dx, xr.grad.view(-1, Cc)) < 5e-3
    assert rel(dy, yr.grad.view(-1, Cc)) < 5e-3


def test_cross_entropy_rows(ops):
    R, W = 999, 101
    logits = torch.randn(R, W) * 3
    logits[5, 7] = float("-inf")
    logits[11, 0] = 50.0
    out3, dl = ops.ce_rows(dev(logits))
    lr = logits.clone().requires_grad_(True)
    loss = F.cross_entropy(lr, torch.zeros(R, dtype=torch.long), reduction="sum")
    loss.backward()
    assert abs(float(out3[0]) - float(loss)) / float(loss) < 1e-5
    mx = logits.argmax(-1) == 0
    mn = logits.argmin(-1) == 0
    assert int(out3[1]) == int(mx.sum()) and int(out3[2]) == int((mx & mn).sum())
    assert float((dl.cpu() - lr.grad).abs().max()) < 1e-5


def test_gather_scatter_rows(ops):
    R, Cc, S = 100, 512, 300
    src = rnd(S, Cc, seed=1)
    idx = torch.randperm(S)[:R].int()
    out = ops.gather_rows(dev(src), dev(idx), R)
    assert torch.equal(out.cpu(), src[idx.long()])
    back = torch.zeros(S, Cc, device="cuda", dtype=BF)
    ops.gather_rows(out, dev(idx), R, scatter=True, out=back)
    want = torch.zeros(S, Cc, dtype=BF)
    want[idx.long()] = src[idx.long()]
    assert torch.equal(back.cpu(), want)


def test_rejects_bad_arguments(ops):
    from wav2vec_s_amd._lib import W2vsError
    with pytest.raises(W2vsError):
        ops.linear_fwd(torch.zeros(4, 8, dtype=BF), torch.zeros(4, 8, dtype=BF).cuda())  # CPU tensor
    with pytest.raises(W2vsError):
        ops.attn_fwd(torch.zeros(1, 10, 3 * 32, dtype=BF, device="cuda"), 1, 10, 4, 0)  # head_dim 32
    with pytest.raises(W2vsError):
        ops.ln_fwd(torch.zeros(4, 2048, dtype=BF, device="cuda"), torch.zeros(2048, dtype=BF, device="cuda"),
                   torch.zeros(2048, dtype=BF, device="cuda"))


def test_fused_adam_matches_fairseq_formula(ops):
    """fs/optim/adam.py:205-229 on an fp32 master with a bf16 working copy."""
    n = 4096 * 3
    g = torch.Generator().manual_seed(0)
    p = torch.randn(n, generator=g)
    grad = torch.randn(n, generator=g) * 3
    lr, b1, b2, eps, wd, scale = 5e-4, 0.9, 0.98, 1e-6, 0.01, 0.25
    p32, m, v = p.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    p16 = torch.zeros(n, dtype=BF).cuda()
    pr, mr, vr = p.clone(), torch.zeros(n), torch.zeros(n)
    for step in (1, 2, 3):
        ops.adam_step(p32, p16, m, v, grad.cuda(), lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, step=step,
                      scale_host=scale)
        gs = grad * scale
        mr.mul_(b1).add_(gs, alpha=1 - b1)
        vr.mul_(b2).addcmul_(gs, gs, value=1 - b2)
        denom = vr.sqrt().add_(eps)
        step_size = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        pr.add_(pr, alpha=-wd * lr)
        pr.addcdiv_(mr, denom, value=-step_size)
    assert float((p32.cpu() - pr).abs().max()) < 1e-6
    assert torch.equal(p16.cpu(), pr.to(BF)) or rel(p16, pr) < 4e-3
    out = torch.zeros(1, device="cuda")
    ops.sumsq(grad.cuda(), out)
    assert abs(float(out) - float((grad ** 2).sum())) / float((grad ** 2).sum()) < 1e-5
    # the norm is reproducible to the bit (partials summed in index order by the last block, no float atomics): data-parallel
    # replicas that hold the same summed gradient must derive the same clip coefficient.  Also: accumulation into out, odd n
    big = torch.randn(5_000_003, generator=torch.Generator().manual_seed(3)).cuda()
    vals = []
    for _ in range(6):
        o = torch.zeros(1, device="cuda")
        ops.sumsq(big[:5_000_000], o)
        vals.append(float(o))
    assert len(set(vals)) == 1, vals
    assert abs(vals[0] - float((big[:5_000_000].double() ** 2).sum())) / vals[0] < 1e-5
    o = torch.full((1,), 2.5, device="cuda")
    ops.sumsq(big[:1_000_003], o)
    ops.sumsq(big[:1_000_003], o)
    want = 2.5 + 2 * float((big[:1_000_003].double() ** 2).sum())
    assert abs(float(o) - want) / want < 1e-5


@pytest.mark.parametrize("clip", [25.0, 0.0])
def test_adam_clip_kernels_follow_the_reference_trajectory(ops, clip):
    """Row f2 pinned: w2vs_sumsq + w2vs_clip_scale_acc + w2vs_adam_step against tests/golden/optim.npz, the trajectory recorded
    from the reference's own Adam (fs/optim/adam.py:103-229), clip_grad_norm_ (fs/utils.py:341-386) and polynomial-decay
    schedule under fs/trainer.py's sequencing (tests/golden/gen_golden_optim.py): 7 updates over warm-up, decay and the floor,
    two of them clipped (clip 25), one with an Inf gradient that must leave master, moments and bf16 image untouched."""
    from conftest import GOLDEN
    fx = np.load(os.path.join(GOLDEN, "optim.npz"))
    tag = "clip%d" % int(clip)
    b1, b2, eps, wd = [float(x) for x in fx["hyper"][:4]]
    n = int(fx["n"])
    p32 = torch.from_numpy(fx["p0"].copy()).cuda()
    p16, m, v = p32.to(BF), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    norm_buf, out3, bad = torch.zeros(1).cuda(), torch.zeros(3).cuda(), torch.zeros(1).cuda()
    num_updates, n_skipped = 0, 0
    for u in range(fx["grads"].shape[0]):
        g = torch.from_numpy(fx["grads"][u].copy()).cuda()
        before = [t.clone() for t in (p32, p16, m, v)]
        ops.sumsq(g, norm_buf)
        ops.clip_scale_acc(norm_buf, out3, bad, scale_host=1.0 / float(fx["sample_size"][u]), clip=clip)
        assert float(norm_buf) == 0.0                                   # consumed: ready for the next update
        ops.adam_step(p32, p16, m, v, g, lr=float(fx[tag + ".lr"][u]), beta1=b1, beta2=b2, eps=eps, weight_decay=wd,
                      step=num_updates + 1, scale_host=1.0, scale_dev=out3[0:1])
        if bool(fx[tag + ".skipped"][u]):
            n_skipped += 1
            assert float(out3[2]) == 1.0 and float(out3[0]) == 0.0 and not np.isfinite(float(out3[1]))
            for a, b in zip(before, (p32, p16, m, v)):
                assert torch.equal(a, b)
        else:
            num_updates += 1
            assert float(out3[2]) == 0.0
            assert abs(float(out3[1]) - float(fx[tag + ".gnorm"][u])) <= 2e-6 * float(fx[tag + ".gnorm"][u]), u
        assert num_updates == int(fx[tag + ".num_updates"][u])
        assert float(bad) == n_skipped
        for name, t in (("p32", p32), ("m", m), ("v", v)):
            want = torch.from_numpy(fx[f"{tag}.{name}"][u]).cuda()
            assert float((t - want).abs().max()) <= 3e-6 * float(want.abs().max()) + 1e-12, (u, name)
        assert torch.equal(p16, p32.to(BF))                              # _sync_fp32_params_to_fp16 (fp16_optimizer.py:218)
    assert n_skipped == 1 and num_updates == 6


def test_transpose_multi(ops):
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072), (40, 24), (65, 130)]
    xs = [dev(rnd(r, c, seed=20 + i)) for i, (r, c) in enumerate(shapes)]
    outs = [torch.empty(c, r, device="cuda", dtype=BF) for (r, c) in shapes]
    ops.transpose_multi([(x.data_ptr(), o.data_ptr(), r, c) for x, o, (r, c) in zip(xs, outs, shapes)])
    torch.cuda.synchronize()
    for x, o in zip(xs, outs):
        assert torch.equal(o, x.t().contiguous())      # bit-exact data movement


# ------------------------------------------------------------------------------------------ GEMM kernel variants
NT_VARIANTS = [(0, 0), (1, 0), (2, 0), (3, 256), (3, 192), (3, 160), (3, 64), (5, 256), (5, 192), (5, 160),
               (3, 1160), (5, 1160), (8, 256), (8, 320), (-1, 0)]     # 1160 = the 160 x 256 tile (round 2); 8 = the 8-phase
                                                                        # kernel (round 3; K % 64 == 0); (-1, 0) = automatic choice


@pytest.mark.parametrize("mode,height", NT_VARIANTS)
def test_gemm_nt_every_variant(ops, mode, height):
    """Each NT kernel (128x128 register / single-buffer / LDS-DMA, loader-consumer and persistent at three tile
    heights) against fp32 torch on ragged shapes with every epilogue, then the conv-as-GEMM form (overlapping A
    rows, batch, the 'row -1' dgrad operand)."""
    try:
        ops.gemm_tune(nt_mode=mode, lc_height=height)
        shapes = [(1000, 392, 200), (257, 128, 64), (6544, 768, 768), (130, 2304, 776), (1700, 3072, 136)]
        if mode == 8:     # whole K tiles only: 1, 2, 3, 5 (odd: padded to a pair), 12 of them; ragged M / N; several tiles per workgroup
            shapes = [(1000, 384, 192), (257, 128, 64), (6544, 768, 768), (130, 2304, 128), (1700, 3072, 320), (70000, 512, 64)]
        for (M, N, K) in shapes:
            x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3)
            y = ops.linear_fwd(dev(x), dev(w), dev(b))
            ref = x.float() @ w.float().t() + b.float()
            assert rel(y, ref) < 5e-3, (M, N, K)
            h, pre = ops.linear_fwd(dev(x), dev(w), dev(b), gelu=True, save_pre=True)
            assert rel(pre, ref) < 5e-3
            assert rel(h, F.gelu(pre.float().cpu())) < 5e-3
            dy, aux = rnd(M, N, seed=4), rnd(M, K, seed=5)
            dx = ops.linear_dgrad(dev(dy), dev(w).t().contiguous())
            assert rel(dx, dy.float() @ w.float()) < 5e-3
            dx2 = ops.linear_dgrad(dev(dy), dev(w).t().contiguous(), dgelu_aux=dev(aux))
            a = aux.float().requires_grad_(True)
            F.gelu(a).backward(dx.float().cpu())
            assert rel(dx2, a.grad) < 6e-3
            dx3 = ops.linear_dgrad(dev(dy), dev(w).t().contiguous(), add_aux=dev(aux))
            assert rel(dx3, dx.float().cpu() + aux.float()) < 5e-3
            # the pair the training step uses: forward saves gelu'(pre), backward multiplies with it
            h2, gp = ops.linear_fwd(dev(x), dev(w), dev(b), gelu=True, save_pre=True, save_grad=True)
            assert torch.equal(h2, h)
            pr = pre.float().cpu().requires_grad_(True)
            F.gelu(pr).sum().backward()
            assert rel(gp, pr.grad) < 5e-3
            dx4 = ops.linear_dgrad(dev(dy), dev(w).t().contiguous(), mul_aux=dev(aux))
            assert rel(dx4, dx.float().cpu() * aux.float()) < 5e-3
        B, Lin, Cin, Cout, k, s = 2, 301, 64, 128, 3, 2
        xx, ww = rnd(B, Lin, Cin, seed=6), rnd(Cout, Cin, k, seed=7, scale=0.1)
        w2 = ops.conv_pack_weight(dev(ww))
        yy, pre = ops.conv_cl_fwd(dev(xx), w2, k, s, None, gelu=True, save_pre=True)
        ref_pre = F.conv1d(xx.float().transpose(1, 2), ww.float(), stride=s).transpose(1, 2)
        assert rel(pre, ref_pre) < 5e-3
        dyy = rnd(*ref_pre.shape, seed=8)
        dxx = ops.conv_cl_dgrad(dev(dyy), w2, k, s, Lin)
        xr = xx.float().requires_grad_(True)
        F.conv1d(xr.transpose(1, 2), ww.float(), stride=s).backward(dyy.float().transpose(1, 2))
        assert rel(dxx, xr.grad) < 5e-3
    finally:
        ops.gemm_tune()


@pytest.mark.parametrize("tn_lc", [0, 1])
def test_gemm_tn_both_kernels(ops, tn_lc):
    """Weight-gradient kernels (128x128 + atomics, loader-consumer + partial-tile workspace) incl. the fused bias
    gradient, accumulation into a non-zero target, ragged K, and the batched conv form."""
    try:
        ops.gemm_tune(tn_lc=tn_lc)
        for (R, N, K) in [(1000, 256, 192), (6544, 768, 768), (333, 640, 512), (4097, 3072, 768)]:
            dy, x = rnd(R, N, seed=1), rnd(R, K, seed=2)
            dw = torch.full((N, K), 0.5, device="cuda")
            db = torch.zeros(N, device="cuda")
            ops.linear_wgrad(dev(dy), dev(x), dw, 1.0, db)
            assert rel(dw, 0.5 + dy.float().t() @ x.float()) < 3e-3, (R, N, K)
            assert rel(db, dy.float().sum(0)) < 3e-3
        B, Lin, Cin, Cout, k, s = 3, 1001, 64, 128, 3, 2
        xx, ww = rnd(B, Lin, Cin, seed=6), rnd(Cout, Cin, k, seed=7, scale=0.1)
        xr, wr = xx.float(), ww.float().requires_grad_(True)
        out = F.conv1d(xr.transpose(1, 2), wr, stride=s)
        dyy = rnd(B, out.shape[-1], Cout, seed=8)
        out.backward(dyy.float().transpose(1, 2))
        dw2 = torch.zeros(Cout, k * Cin, device="cuda")
        ops.conv_cl_wgrad(dev(dyy), dev(xx), k, s, dw2)
        assert rel(dw2, wr.grad.permute(0, 2, 1).reshape(Cout, k * Cin)) < 3e-3
    finally:
        ops.gemm_tune()


def test_conv0_ln_gelu_with_conv_bias(ops):
    """The matrix-core conv0 carries the conv bias as an extra all-ones tap: forward and every gradient
    (weight, conv bias, LayerNorm affine) against autograd, frames straddling utterance ends."""
    B, L, Cc, k, s = 3, 2003, 512, 10, 5
    wave = rnd(B, L, seed=11)
    w = rnd(Cc, 1, k, seed=12, scale=0.4)
    cb = rnd(Cc, seed=13, scale=0.3)
    g, b = (1 + 0.1 * torch.randn(Cc)).to(BF), (0.1 * torch.randn(Cc)).to(BF)
    y, mean, rstd = ops.conv0_fwd(dev(wave), dev(w), dev(g), dev(b), k, s, conv_bias=dev(cb))
    wf, cbf, gf, bf_ = (t.float().requires_grad_(True) for t in (w, cb, g, b))
    c = F.conv1d(wave.float().unsqueeze(1), wf, cbf, stride=s).transpose(1, 2)          # [B, L0, C]
    ref = F.gelu(F.layer_norm(c, (Cc,), gf, bf_))
    assert rel(y, ref) < 5e-3
    dy = rnd(*y.shape, seed=14)
    ref.backward(dy.float())
    dw = torch.zeros(Cc, k, device="cuda")
    dcb = torch.zeros(Cc, device="cuda")
    dg = torch.zeros(Cc, device="cuda")
    db = torch.zeros(Cc, device="cuda")
    ops.conv0_bwd(dev(wave), dev(w), dev(g), dev(b), mean, rstd, dev(dy), k, s, dw, dg, db, conv_bias=dev(cb),
                  dconv_bias=dcb)
    assert rel(dw, wf.grad.view(Cc, k)) < 5e-3
    assert rel(dg, gf.grad) < 5e-3
    assert rel(db, bf_.grad) < 5e-3
    # d(conv bias) is a sum of the LayerNorm-backward output, which is zero-mean per frame: compare with an absolute floor
    assert float((dcb.cpu() - cbf.grad).abs().max()) < 5e-3 * max(1.0, float(cbf.grad.abs().max())) + 2e-2


@pytest.mark.parametrize("mode,height", [(8, 256), (8, 320), (5, 256), (5, 160), (5, 1160), (-1, 0)])
def test_gemm_nt_structural_zero_block_is_skipped_exactly(ops, mode, height):
    """w2vs_gemm_desc.zk_col / zk_k: B[n][k] == 0 for n >= zk_col, k < zk_k (the [[W2, 0], [W0, W1]] operand of the (3,2)-conv
    input gradient).  The persistent kernels start the K loop of the tiles at those columns at zk_k - a quarter of the
    multiply-adds gone; adding exact zeros changes no fp32 sum, so the result must be BIT-identical to the plain product, over
    several tiles per workgroup, batch planes and a ragged M.  A (true) promise the tile grid cannot honour - zk_col inside a tile, an odd number
    of K tiles - is ignored, not mis-applied; a narrower true promise (zk_col 768) skips fewer tiles."""
    Bz, M, N, K = 3, 21000, 1024, 1024          # 66 x 4 x 3 = 792 tiles of 320 x 256: three rounds of 256 workgroups, row groups straddle them
    a = dev(rnd(Bz, M, K, seed=1))
    b = rnd(N, K, seed=2, scale=0.05)
    b[512:, :512] = 0
    b = dev(b)
    aux = dev(rnd(Bz, M, N, seed=3))
    try:
        ops.gemm_tune(nt_mode=mode, lc_height=height)
        outs = []
        for zk in (dict(), dict(zk_col=512, zk_k=512), dict(zk_col=576, zk_k=512), dict(zk_col=512, zk_k=448), dict(zk_col=768, zk_k=512)):
            out = torch.empty(Bz, M, N, device="cuda", dtype=BF)
            ops.gemm_nt(a, b, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, out=out, aux=aux, epi=8, batch=Bz, sA=M * K, sC=M * N, **zk)
            outs.append(out)
        ref = (a[1].float() @ b.float().t()) * aux[1].float()
        assert rel(outs[0][1], ref) < 5e-3
        for o in outs[1:]:
            assert torch.equal(o, outs[0])
    finally:
        ops.gemm_tune()


def test_gemm_tn_group_equals_single_launches():
    """w2vs_gemm_tn_group: the four weight gradients of an encoder layer (shared token dimension) in one launch, no K
    split, single writer per tile - same result as four w2vs_gemm_tn calls, accumulation into non-zero targets and the
    bias column sums included; and the fallback when the group does not qualify."""
    from wav2vec_s_amd import ops
    g = torch.Generator().manual_seed(0)
    R, E, F = 5584, 768, 3072
    mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.5).to(BF).cuda()      # noqa: E731
    dys = [mk(R, E), mk(R, F), mk(R, E), mk(R, 3 * E)]
    xs = [mk(R, F), mk(R, E), mk(R, E), mk(R, E)]

    def targets():
        gg = torch.Generator().manual_seed(1)
        return ([torch.randn(dy.shape[1], x.shape[1], generator=gg).cuda() for dy, x in zip(dys, xs)],
                [torch.randn(dy.shape[1], generator=gg).cuda() for dy in dys])
    w1, b1 = targets()
    for dy, x, w, b in zip(dys, xs, w1, b1):
        ops.linear_wgrad(dy, x, w, 0.5, b)
    w2, b2 = targets()
    ops.gemm_tn_group([dict(a=dy, b=x, out_f32=w, M=dy.shape[1], N=x.shape[1], K=R, lda=dy.shape[1], ldb=x.shape[1],
                            ldc=x.shape[1], alpha=0.5, colsum_out=b) for dy, x, w, b in zip(dys, xs, w2, b2)])
    for a, b in zip(w1 + b1, w2 + b2):
        assert rel(b, a) < 2e-5, rel(b, a)            # same products, different (K-split) summation order
    ref = (dys[1].float().t() @ xs[1].float()) * 0.5
    assert rel(w2[1] - targets()[0][1], ref) < 2e-3
    # a group that does not fill half the chip runs as single launches (the same kernels linear_wgrad uses)
    w3, b3 = targets()
    ops.gemm_tn_group([dict(a=dys[2], b=xs[2], out_f32=w3[2], M=E, N=E, K=R, lda=E, ldb=E, ldc=E, alpha=0.5, colsum_out=b3[2])])
    assert torch.equal(w3[2], w1[2]) and rel(b3[2], b1[2]) < 1e-5     # the column sums arrive through atomics: order varies


def test_gemm_tn_group_pair_split_needs_co_residency():
    """The 8-phase grouped launch splits K over workgroup PAIRS that wait on each other's flag: that form (14) is taken only
    when the whole grid is co-resident on THIS device (queried occupancy x CU count; a hint can only lower it) and the caller
    has not forbidden it (w2vs_gemm_tn8_max_split(1): what the training step sets while an all-reduce may hold CUs).  Every
    fallback computes the same sums."""
    from wav2vec_s_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    R, E, F = 5584, 768, 3072
    mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.5).to(BF).cuda()      # noqa: E731
    dys = [mk(R, E), mk(R, F), mk(R, E), mk(R, 3 * E)]
    xs = [mk(R, F), mk(R, E), mk(R, E), mk(R, E)]

    def run(num_cu, max_split):
        lib.w2vs_gemm_tn8_max_split(max_split)
        try:
            ws = [torch.zeros(dy.shape[1], x.shape[1], device="cuda") for dy, x in zip(dys, xs)]
            bs = [torch.zeros(dy.shape[1], device="cuda") for dy in dys]
            ops.gemm_tn_group([dict(a=dy, b=x, out_f32=w, M=dy.shape[1], N=x.shape[1], K=R, lda=dy.shape[1], ldb=x.shape[1],
                                    ldc=x.shape[1], colsum_out=b) for dy, x, w, b in zip(dys, xs, ws, bs)], num_cu=num_cu)
            torch.cuda.synchronize()
            return lib.w2vs_gemm_last_group_form(), ws, bs
        finally:
            lib.w2vs_gemm_tn8_max_split(2)
    def run_overwrite(num_cu, max_split):
        """The same launch with w2vs_gemm_desc.overwrite = 1 into NaN-filled targets: C = A^T B, the bias sums still accumulate."""
        lib.w2vs_gemm_tn8_max_split(max_split)
        try:
            ws = [torch.full((dy.shape[1], x.shape[1]), float("nan"), device="cuda") for dy, x in zip(dys, xs)]
            bs = [torch.zeros(dy.shape[1], device="cuda") for dy in dys]
            ops.gemm_tn_group([dict(a=dy, b=x, out_f32=w, M=dy.shape[1], N=x.shape[1], K=R, lda=dy.shape[1], ldb=x.shape[1],
                                    ldc=x.shape[1], colsum_out=b, overwrite=1) for dy, x, w, b in zip(dys, xs, ws, bs)], num_cu=num_cu)
            torch.cuda.synchronize()
            return lib.w2vs_gemm_last_group_form(), ws, bs
        finally:
            lib.w2vs_gemm_tn8_max_split(2)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    form, w0, b0 = run(cus, 2)
    assert form == 14 if cus >= 216 else form != 14            # 108 tiles of 256^2 x 2: needs 216 co-resident workgroups
    for num_cu, max_split, forms in ((cus, 1, (0, 12)), (215, 2, (0, 12)), (100, 2, (0,)), (1 << 20, 2, (form,))):
        f_, w_, b_ = run(num_cu, max_split)
        assert f_ in forms, (num_cu, max_split, f_)
        for a, b in zip(w0 + b0, w_ + b_):
            assert rel(b, a) < 2e-5, (num_cu, max_split, rel(b, a))
        fo, wo, bo = run_overwrite(num_cu, max_split)      # every form: the overwriting launch equals the accumulation into zeros
        assert fo == f_
        for a, b in zip(w0 + b0, wo + bo):
            assert not bool(torch.isnan(b).any()) and rel(b, a) < 2e-5, (num_cu, max_split, "overwrite")


def test_group_attention_beyond_one_record_table(ops):
    """Cross (joiner) mode with more query tiles than one launch's record table holds (G*U = 8 712 rows = 273 tiles > 256:
    the launch is chunked) against plain torch, forward and backward; and the limit itself (512 tiles) is an error, not an
    overrun (rain/layers/attention_transducer.py:591-716 is what this mode replaces)."""
    torch.manual_seed(11)
    B, H, G, U, m, C = 1, 2, 33, 264, 4, 128
    S, Nq = G * m, G * U
    q = (torch.randn(B, Nq, C) * 0.5).to(BF)
    kv = (torch.randn(B, S, 2 * C) * 0.5).to(BF)
    do = torch.randn(B, Nq, C).to(BF)
    o, lse = ops.group_attn_fwd(dev(q), dev(kv), H, m, U)
    dq, dkv = ops.group_attn_bwd(dev(do), dev(q), dev(kv), o, lse, H, m, U)
    qr, kvr = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    qh = qr.view(B, Nq, H, C // H).transpose(1, 2)
    kh = kvr[..., :C].reshape(B, S, H, C // H).transpose(1, 2)
    vh = kvr[..., C:].reshape(B, S, H, C // H).transpose(1, 2)
    sc = qh @ kh.transpose(-1, -2) * (C // H) ** -0.5
    vis = torch.arange(S).view(1, S) < ((torch.arange(Nq) // U + 1) * m).clamp(max=S).view(Nq, 1)
    sc = sc.masked_fill(~vis, float("-inf"))
    ref = (sc.softmax(-1) @ vh).transpose(1, 2).reshape(B, Nq, C)
    ref.backward(do.double())
    assert rel(o, ref.detach()) < 1e-2
    assert rel(dq, qr.grad) < 2e-2 and rel(dkv, kvr.grad) < 2e-2
    from wav2vec_s_amd._lib import W2vsError
    big = torch.zeros(1, 16416, C, dtype=BF, device="cuda")          # 513 tiles
    with pytest.raises(W2vsError):
        ops.group_attn_fwd(big, dev(kv), H, m, 16416 // 4)
