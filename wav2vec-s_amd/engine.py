"""Forward / backward schedule of one wav2vec-S pre-training micro-step on one MI355X.

This is the hand-written replacement of autograd over the reference's forward
(fs/models/wav2vec/wav2vec2.py:544-658 + wav2vec_S.py:355-440): an explicit sequence of
libw2vs kernel launches on the current HIP stream, with every saved activation chosen by hand
(bf16, channel-last / token-major) and all parameter gradients accumulated in ONE flat fp32
arena (the buffer the data-parallel all-reduce later works on).  No torch math op is used on
the path; torch allocates buffers and carries views.
"""
import os
from typing import Dict, Optional

import numpy as np
import torch

import ctypes as C

from . import _lib, host_rng, ops
from ._lib import LayerDesc, W2vsError

BF16 = torch.bfloat16
_GOLD = 0x9E3779B97F4A7C15
_MASK64 = (1 << 64) - 1


def _site_seed(base: int, site: int) -> int:
    x = (base + site * _GOLD) & _MASK64
    x ^= x >> 31
    x = (x * 0xBF58476D1CE4E5B9) & _MASK64
    x ^= x >> 29
    return x & _MASK64


class Draws:
    """All host-side random decisions of one forward, injectable for parity runs."""

    def __init__(self, mask_indices=None, neg_idx=None, context=None, layer_keep=None, gumbel_noise=None):
        self.mask_indices = mask_indices   # np.bool [B, T] or None (-> sampled)
        self.neg_idx = neg_idx             # torch int64 [B, K*M] (CPU) or None
        self.context = context             # (m, r) or None
        self.layer_keep = layer_keep       # list[bool] or None
        self.gumbel_noise = gumbel_noise   # torch fp32 [B*M*G, V] (CPU/GPU) or None (-> device RNG)


class Arena:
    """Flat fp32 gradient arena with named views."""

    def __init__(self, shapes: Dict[str, tuple], device, flat: Optional[torch.Tensor] = None, layout=None):
        if layout is not None:               # (offsets, numel) of an earlier arena with the same shapes: nothing to recompute
            self.offsets, self.numel = layout
        else:
            self.offsets = {}
            off = 0
            for n, shp in shapes.items():
                numel = int(np.prod(shp))
                self.offsets[n] = (off, numel, shp)
                off += (numel + 3) // 4 * 4  # keep 16-B alignment of every view
            self.numel = off
        self.flat = torch.zeros(self.numel, device=device, dtype=torch.float32) if flat is None else flat

    def view(self, n):
        off, numel, shp = self.offsets[n]
        return self.flat[off:off + numel].view(shp)

    def __contains__(self, n):
        return n in self.offsets


class State:
    pass


def _upload(arrays: Dict[str, np.ndarray], dev) -> Dict[str, torch.Tensor]:
    """Every host-made index array of a step travels in ONE pinned buffer with ONE async copy, issued
    before the first kernel launch, so nothing in the step blocks the host on the GPU."""
    metas, off = [], 0
    for n, a in arrays.items():
        a = np.ascontiguousarray(a)
        metas.append((n, a, off))
        off += (a.nbytes + 15) // 16 * 16
    host = torch.empty(max(off, 16), dtype=torch.uint8, pin_memory=True)
    hv = host.numpy()
    for n, a, o in metas:
        hv[o:o + a.nbytes] = a.view(np.uint8).reshape(-1)
    devbuf = host.to(dev, non_blocking=True)
    out = {"_host": host, "_dev": devbuf}
    tmap = {np.dtype("uint8"): torch.uint8, np.dtype("int32"): torch.int32, np.dtype("int64"): torch.int64,
            np.dtype("bool"): torch.uint8}
    for n, a, o in metas:
        t = devbuf[o:o + a.nbytes].view(tmap[a.dtype]).view(a.shape)
        out[n] = t
    return out


def _pname(i, tail):
    return f"feature_extractor.conv_layers.{i}.{tail}"


def forward(cfg, W: Dict[str, torch.Tensor], source: torch.Tensor, *, training: bool, mask: bool,
            features_only: bool, padding_mask: Optional[torch.Tensor], draws: Draws, rng_base: int,
            tau: float, packed: Optional[Dict[str, torch.Tensor]] = None, upload_cache: Optional[dict] = None,
            need_backward: bool = True) -> State:
    """W: name -> bf16 contiguous device tensor (reference state_dict names).  source [B, L] bf16.
    packed: conv weights already tap-major [Cout, k*Cin] (flat parameter storage), else packed here.
    upload_cache: a dict the caller keeps (the streaming twin): without a mask and without padding every index array of the
    step is a function of the shape alone, so its device copy is made once per (B, L, m, r) and reused - a call then contains
    no host-to-device copy at all (which is also what lets it be captured into a HIP graph).
    need_backward = False (no input of the call requires a gradient): the layers skip what only a backward would read."""
    st = State()
    packed = packed or {}
    st.packed = {}
    st.cfg, st.W, st.training, st.features_only = cfg, W, training, features_only
    dev = source.device
    B, L = source.shape
    convs = cfg.conv_layers
    group_norm = cfg.extractor_mode == "default"
    if cfg.activation_dropout != 0.0 and training:
        raise W2vsError("activation_dropout > 0 is not built (wav2vec-S configs use 0.0)")
    if cfg.quantize_input or cfg.target_glu or cfg.negatives_from_everywhere or cfg.cross_sample_negatives \
            or cfg.codebook_negatives or cfg.mask_channel_prob > 0:
        raise W2vsError("quantize_input / target_glu / negatives_from_everywhere / cross_sample / codebook "
                        "negatives / channel masks are not part of the wav2vec-S hot path and are not built")
    p_in = cfg.dropout_input if training else 0.0
    p_feat = cfg.dropout_features if training else 0.0
    p_enc = cfg.dropout if training else 0.0
    p_att = cfg.attention_dropout if training else 0.0
    seed = lambda site: _site_seed(rng_base, site)  # noqa: E731
    st.seed = seed
    st.p = (p_in, p_feat, p_enc, p_att)

    # ------------------------------------------------------------------ host phase (a6, a9, a10, a16, a21)
    # T follows from L alone, so every host draw (mask -> context -> LayerDrop -> negatives, each from
    # its own generator in the reference's order) and every index array is made BEFORE the first launch.
    convs_ = cfg.conv_layers
    T = L
    for (_, k_, s_) in convs_:
        T = (T - k_) // s_ + 1
    C0 = convs_[-1][0]
    E = cfg.encoder_embed_dim
    pad_frames = None
    if padding_mask is not None:   # padding mask -> frame resolution (wav2vec2.py:560-565)
        pm = padding_mask.bool().cpu()
        extra = pm.size(1) % T
        if extra > 0:
            pm = pm[:, :-extra]
        pad_frames = pm.view(pm.size(0), T, -1).all(-1)
    st.pad_frames = pad_frames
    mask_np = None
    if mask and cfg.mask_prob > 0:
        if draws.mask_indices is not None:
            mask_np = np.asarray(draws.mask_indices, dtype=bool)
        else:
            mask_np = host_rng.compute_mask_indices((B, T), pad_frames, cfg.mask_prob, cfg.mask_length,
                                                    cfg.mask_selection, cfg.mask_other, min_masks=2,
                                                    no_overlap=cfg.no_mask_overlap, min_space=cfg.mask_min_space)
    st.mask_np = mask_np
    mult = cfg.required_seq_len_multiple
    Tp = T + ((-T) % mult)
    m_ctx, r_ctx = draws.context if draws.context is not None else host_rng.sample_context(
        cfg.context_type, cfg.main_context, cfg.right_context)
    lay = host_rng.block_layout(Tp, m_ctx, r_ctx)
    N = lay.N
    st.Tp, st.N, st.lay, st.m, st.r = Tp, N, lay, m_ctx, r_ctx
    keep = draws.layer_keep if draws.layer_keep is not None else host_rng.layerdrop_keep(
        cfg.encoder_layers, cfg.encoder_layerdrop, training)
    st.kept = [i for i in range(cfg.encoder_layers) if keep[i]]
    pad_np = None
    if pad_frames is not None or Tp != T:
        pad_np = np.zeros((B, Tp), dtype=bool)
        if pad_frames is not None:
            pad_np[:, :T] = pad_frames.numpy()
        pad_np[:, T:] = True               # wav2vec_S.py:378-384
    kpad_np = lay.key_padding(pad_np, B)
    up = {"pos": host_rng.positions_from_padding(pad_frames, B, T).numpy(), "src": lay.src,
          "copy_start": lay.copy_start, "copy_list": lay.copy_list}
    if mask_np is not None:
        up["mask"] = mask_np.astype(np.uint8)
    if kpad_np is not None:
        up["kpad"] = kpad_np
    if pad_frames is not None:
        up["pad"] = pad_frames.numpy().astype(np.uint8)
    if features_only:
        up["out_idx"] = (np.arange(B)[:, None] * N + np.arange(T)[None, :]).reshape(-1).astype(np.int32)
    else:
        if mask_np is None:
            raise W2vsError("the pre-training head needs mask=True and mask_prob > 0")
        if not cfg.quantize_targets:
            raise W2vsError("quantize_targets=False is not built (wav2vec-S pre-training quantizes targets)")
        bidx, tidx = np.nonzero(mask_np)
        M = len(tidx) // B
        st.M, st.K = M, cfg.num_negatives
        up["frame_idx"] = (bidx * T + tidx).astype(np.int32)      # rows of [B*T, .]
        up["token_idx"] = (bidx * N + tidx).astype(np.int32)      # rows of [B*N, .]
        neg = draws.neg_idx if draws.neg_idx is not None else host_rng.sample_negative_indices(B, M, st.K)
        up["neg"] = neg.numpy() if torch.is_tensor(neg) and not neg.is_cuda else None
        if up["neg"] is None:
            del up["neg"]
            st.neg = neg
    ckey = None
    if upload_cache is not None and mask_np is None and pad_frames is None and features_only:
        ckey = (B, L, m_ctx, r_ctx, mult, str(dev))
    if ckey is not None and ckey in upload_cache:
        st.up = upload_cache[ckey]
    else:
        st.up = _upload(up, dev)
        if ckey is not None:
            if len(upload_cache) >= 256:
                upload_cache.pop(next(iter(upload_cache)))
            upload_cache[ckey] = st.up
    if "neg" in st.up:
        st.neg = st.up["neg"]
    mask_dev = st.up.get("mask")
    st.mask_dev = mask_dev
    st.kpad = st.up.get("kpad")
    st.pad_dev = st.up.get("pad")
    pos = st.up["pos"]
    st.pos = pos
    st.src, st.copy_start, st.copy_list = st.up["src"], st.up["copy_start"], st.up["copy_list"]

    # ------------------------------------------------------------------ feature extractor (a1)
    ln_num = cfg.layer_norm_num
    st.conv = []
    dim0, k0, s0 = convs[0]
    w0 = packed.get(_pname(0, "0.weight"))
    if w0 is None:
        w0 = W[_pname(0, "0.weight")]
    st.packed[0] = w0                      # Cin == 1: [C, 1, k] and [C, k*1] are the same bytes
    st.source = source
    if group_norm:   # Fp32GroupNorm(C, C) on layer 0 only (wav2vec2.py:744-750, 767)
        y, gstat = ops.conv0_gn_fwd(source, w0, W[_pname(0, "2.weight")], W[_pname(0, "2.bias")], k0, s0,
                                    conv_bias=W.get(_pname(0, "0.bias")))
        st.conv.append(dict(y=y, gstat=gstat))
    else:
        y, mean, rstd = ops.conv0_fwd(source, w0, W[_pname(0, "2.1.weight")],
                                      W[_pname(0, "2.1.bias")], k0, s0, conv_bias=W.get(_pname(0, "0.bias")))
        st.conv.append(dict(y=y, mean=mean, rstd=rstd))
    x = y
    for i in range(1, len(convs)):
        dim, k, s = convs[i]
        w2 = packed.get(_pname(i, "0.weight"))
        if w2 is None:
            w2 = ops.conv_pack_weight(W[_pname(i, "0.weight")])
        st.packed[i] = w2
        bias = W.get(_pname(i, "0.bias"))
        rec = dict(x_in=x, k=k, s=s, ln=(not group_norm and i < ln_num))
        if rec["ln"]:
            c = ops.conv_cl_fwd(x, w2, k, s, bias, gelu=False, save_pre=False)
            y, _, mean, rstd = ops.ln_fwd(c, W[_pname(i, "2.1.weight")], W[_pname(i, "2.1.bias")], gelu=True)
            rec.update(c=c, mean=mean, rstd=rstd)
        else:
            # layers below the last save gelu'(pre) (the next layer's dgrad multiplies with it); the last one saves
            # pre itself, which the feature LayerNorm backward differentiates through
            y, pre = ops.conv_cl_fwd(x, w2, k, s, bias, gelu=True, save_pre=True, save_grad=(i < len(convs) - 1))
            rec.update(pre=pre)
        st.conv.append(rec)
        x = y
    y_last = x                                   # [B, T, C0] post-GELU features
    assert y_last.shape[1] == T and y_last.shape[2] == C0
    st.B, st.T, st.C0 = B, T, C0

    # ------------------------------------------------------------------ features_pen + LayerNorm (a3, a4)
    st.pen_acc = ops.zeros((4,), torch.float32, dev)[:1]
    feats, _, st.f_mean, st.f_rstd = ops.ln_fwd(y_last, W["layer_norm.weight"], W["layer_norm.bias"], sumsq=st.pen_acc)
    st.y_last, st.feats = y_last, feats

    # ------------------------------------------------------------------ post_extract_proj (a5)
    if "post_extract_proj.weight" in W:
        xproj = ops.linear_fwd(feats.view(B * T, C0), W["post_extract_proj.weight"], W["post_extract_proj.bias"]).view(B, T, E)
    else:
        xproj = feats
    st.xproj = xproj

    # ------------------------------------------------------------------ encoder prologue (a7-a10)
    table = _pos_table(E, dev)
    post_ln = not cfg.layer_norm_first
    x0, st.p_mean, st.p_rstd = ops.enc_prologue_fwd(
        xproj, mask_dev, st.pad_dev, pos, W["mask_emb"], table, W["encoder.layer_norm.weight"],
        W["encoder.layer_norm.bias"], st.src, Tp, apply_ln=post_ln, p_in=p_in, seed_in=seed(1), p_enc=p_enc,
        seed_enc=seed(2))

    # ------------------------------------------------------------------ encoder layers (a11, a12)
    H = cfg.encoder_attention_heads
    R = B * N
    st.layers = []
    x = x0.view(R, E)
    st.x0 = x
    F = cfg.encoder_ffn_embed_dim
    nk = len(st.kept)
    # one bf16 slab and one fp32 slab hold every saved activation of every kept layer
    per16 = R * (8 * E + 2 * F)
    slab16 = ops.empty((max(nk, 1) * per16 + R * E,), BF16, dev)
    # attention-dropout keep masks (w2vs_attn_desc.drop_bits): written by each layer's forward, read by its backward.
    # OFF by default: measured on MI355X at the cfgB shape the backward gains 3 us per layer (131.7 vs 134.6 us: its two
    # kernels are latency-, not hash-bound) while the forward pays 5 us for parking the masks (40.9 vs 36.1 us).
    nbits = B * H * ((N + 31) // 32) ** 2 * 32 if (p_att > 0 and KEEP_MASK_STORE) else 0
    per32 = B * H * N + 4 * R + nbits
    slab32 = ops.empty((max(nk, 1) * per32,), torch.float32, dev)
    st.tmp = slab16[max(nk, 1) * per16:]
    stream = ops._stream()
    kpad_ptr = st.kpad.data_ptr() if st.kpad is not None else None
    use_sel = SELECT_LAST_LAYER and not features_only and "token_idx" in st.up and nk > 0
    names = [f"encoder.layers.{li}." for li in st.kept]
    s_stream = None
    if not post_ln:
        # pre-LN (wav2vec2.py:932-953): x is the residual stream; every "residual add + next LayerNorm" pair is one fused
        # kernel inside the composite layer call, so only the FIRST norm runs on its own
        s_stream = x
        if nk > 0:
            x, _, st.n0_mean, st.n0_rstd = ops.ln_fwd(s_stream, W[names[0] + "self_attn_layer_norm.weight"],
                                                      W[names[0] + "self_attn_layer_norm.bias"])
            st.n0 = x                     # layer 0's descriptor holds the raw pointer: keep the buffer alive for the backward
    for j, li in enumerate(st.kept):
        pre = names[j]
        d = LayerDesc()
        d.B, d.N, d.E, d.F, d.H, d.Tp, d.m, d.r, d.post_ln, d.num_cu = B, N, E, F, H, Tp, m_ctx, r_ctx, int(post_ln), 256
        d.p_drop, d.p_attn = p_enc, p_att
        d.seed_attn, d.seed_drop1, d.seed_drop2 = seed(100 + 4 * li), seed(101 + 4 * li), seed(102 + 4 * li)
        d.kpad = kpad_ptr
        wqkv, bqkv = packed[pre + "qkv"] if (pre + "qkv") in packed else _qkv_pack(W, pre)
        rec = dict(li=li, wqkv=wqkv, bqkv=bqkv)
        d.wqkv, d.bqkv = wqkv.data_ptr(), bqkv.data_ptr()
        if post_ln:
            norm_a = (pre + "self_attn_layer_norm.weight", pre + "self_attn_layer_norm.bias")
            norm_b = (pre + "final_layer_norm.weight", pre + "final_layer_norm.bias")
        else:      # ln1 = the norm between attention and FFN, ln2 = the NEXT norm (include/w2vs.h, pre-LN form)
            norm_a = (pre + "final_layer_norm.weight", pre + "final_layer_norm.bias")
            norm_b = ((names[j + 1] + "self_attn_layer_norm.weight", names[j + 1] + "self_attn_layer_norm.bias")
                      if j + 1 < nk else ("encoder.layer_norm.weight", "encoder.layer_norm.bias"))
        rec["norm_a"], rec["norm_b"] = norm_a, norm_b
        for f_, n_ in (("wo", pre + "self_attn.out_proj.weight"), ("bo", pre + "self_attn.out_proj.bias"),
                       ("ln1_g", norm_a[0]), ("ln1_b", norm_a[1]), ("w1", pre + "fc1.weight"), ("b1", pre + "fc1.bias"),
                       ("w2", pre + "fc2.weight"), ("b2", pre + "fc2.bias"), ("ln2_g", norm_b[0]), ("ln2_b", norm_b[1])):
            setattr(d, f_, W[n_].data_ptr())
        base16 = slab16.data_ptr() + 2 * j * per16
        o = 0
        for f_, cols in (("qkv", 3 * E), ("ctx", E), ("s1", E), ("x1", E), ("hpre", F), ("h", F), ("s2", E), ("x_out", E)):
            setattr(d, f_, base16 + 2 * o)
            o += R * cols
        if not need_backward:
            d.hpre = None                  # inference: fc1's epilogue does not store gelu'(pre)
        base32 = slab32.data_ptr() + 4 * j * per32
        d.lse = base32
        d.mean1, d.rstd1 = base32 + 4 * (B * H * N), base32 + 4 * (B * H * N + R)
        d.mean2, d.rstd2 = base32 + 4 * (B * H * N + 2 * R), base32 + 4 * (B * H * N + 3 * R)
        d.drop_bits = base32 + 4 * (B * H * N + 4 * R) if nbits else None
        d.x_in = x.data_ptr()
        if not post_ln:
            d.stream_in = s_stream.data_ptr()
        d.tmp = st.tmp.data_ptr()
        sel_last = use_sel and j == nk - 1
        if sel_last:
            # only the masked frames of the LAST layer's output are ever read (x[mask_indices], wav2vec2.py:590):
            # everything behind its attention runs on those B*M rows, the attention on the T' main frames
            RMs = st.up["token_idx"].numel()
            rec["sel_bufs"] = (ops.empty((RMs, E), BF16, dev), ops.empty((RMs, E), BF16, dev))
            d.sel_idx, d.n_sel, d.n_q = st.up["token_idx"].data_ptr(), RMs, Tp
            d.ctx_sel, d.xin_sel = rec["sel_bufs"][0].data_ptr(), rec["sel_bufs"][1].data_ptr()
        _lib.call("w2vs_layer_fwd", C.byref(d), stream)
        rec["desc"] = d
        st.layers.append(rec)
        rows = RMs if sel_last else R
        x_off = j * per16 + R * (7 * E + 2 * F)
        x = slab16[x_off: x_off + rows * E].view(-1, E)
        if not post_ln:
            s_off = j * per16 + R * (6 * E + 2 * F)
            s_stream = slab16[s_off: s_off + rows * E].view(-1, E)
    st.slabs = (slab16, slab32)
    st.enc_is_sel = bool(st.layers) and use_sel
    if post_ln or nk > 0:
        enc = x                            # pre-LN: the last composite call applied encoder.layer_norm (wav2vec2.py:831-832)
    else:
        enc, _, st.fin_mean, st.fin_rstd = ops.ln_fwd(s_stream, W["encoder.layer_norm.weight"], W["encoder.layer_norm.bias"])
    st.enc = enc                           # [B*N, E]

    if features_only:
        st.out_idx = st.up["out_idx"]
        st.out_x = ops.gather_rows(enc, st.out_idx, B * T).view(B, T, E)
        return st

    # ------------------------------------------------------------------ loss head (a14-a18)
    M, K = st.M, st.K
    frame_idx, token_idx = st.up["frame_idx"], st.up["token_idx"]
    st.frame_idx, st.token_idx = frame_idx, token_idx
    RM = B * M
    y_in = ops.gather_rows(feats.view(B * T, C0), frame_idx, RM)
    if p_feat > 0:
        y_in = ops.dropout(y_in, p_feat, seed(3))
    st.y_in = y_in
    G, V = cfg.latent_groups, cfg.latent_vars
    # fp32 logits (bias added inside the quantizer kernel): the code selection is an argmax and must not see bf16 sums
    st.q_logits = ops.linear_fwd_f32(y_in, W["quantizer.weight_proj.weight"])
    vars2d = W["quantizer.vars"].view(G * V, -1)
    noise = draws.gumbel_noise
    if noise is not None:
        noise = noise.to(dev).float().contiguous()
    st.noise = noise
    st.tau = tau
    q, st.qst = ops.quant_fwd(st.q_logits, vars2d, G, V, tau, training, noise=noise, seed=seed(4),
                              bias=W["quantizer.weight_proj.bias"])
    st.q = q
    yq = ops.linear_fwd(q, W["project_q.weight"], W["project_q.bias"])
    xm = enc if getattr(st, "enc_is_sel", False) else ops.gather_rows(enc, token_idx, RM)
    xf = ops.linear_fwd(xm, W["final_proj.weight"], W["final_proj.bias"])
    st.yq, st.xm, st.xf = yq, xm, xf
    if not st.neg.is_cuda:
        st.neg = st.neg.to(dev)
    st.logits, st.nce_norms = ops.nce_fwd(xf, yq, st.neg, B, M, K, cfg.logit_temp)   # [B*M, K+1] rows (b, m)
    return st


def conv_features(conv_layers, mode, ln_num, W, source):
    """The extractor alone, forward only (ConvFeatureExtractionModel.forward, wav2vec2.py:773-781): source [B, L] bf16 ->
    [B, T, C] channel-last post-GELU features.  Same kernels as the training forward."""
    group_norm = mode == "default"
    dim0, k0, s0 = conv_layers[0]
    w0 = W[_pname(0, "0.weight")]
    if group_norm:
        x, _ = ops.conv0_gn_fwd(source, w0, W[_pname(0, "2.weight")], W[_pname(0, "2.bias")], k0, s0,
                                conv_bias=W.get(_pname(0, "0.bias")))
    else:
        x, _, _ = ops.conv0_fwd(source, w0, W[_pname(0, "2.1.weight")], W[_pname(0, "2.1.bias")], k0, s0,
                                conv_bias=W.get(_pname(0, "0.bias")))
    for i in range(1, len(conv_layers)):
        _, k, s = conv_layers[i]
        w2 = ops.conv_pack_weight(W[_pname(i, "0.weight")])
        bias = W.get(_pname(i, "0.bias"))
        if not group_norm and i < ln_num:
            c = ops.conv_cl_fwd(x, w2, k, s, bias, gelu=False, save_pre=False)
            x, _, _, _ = ops.ln_fwd(c, W[_pname(i, "2.1.weight")], W[_pname(i, "2.1.bias")], gelu=True)
        else:
            x = ops.conv_cl_fwd(x, w2, k, s, bias, gelu=True, save_pre=False)
    return x


def single_layer_forward(layer, x, padding_mask=None):
    """One encoder layer on its own (TransformerSentenceEncoderLayer.forward, wav2vec2.py:921-978): x [T, B, C] ->
    [T, B, C], full attention (one block of T keys), eval semantics.  One w2vs_layer_fwd call."""
    T, B, E = x.shape
    dev = x.device
    dt = x.dtype
    par = {n: (p if p.dtype == BF16 else p.to(BF16)).contiguous() for n, p in layer.named_parameters()}
    H = layer.self_attn.num_heads
    F = layer.fc1.out_features
    idx = (torch.arange(B, device=dev, dtype=torch.int32).view(B, 1) + torch.arange(T, device=dev, dtype=torch.int32).view(1, T) * B)
    xb = ops.gather_rows(x.to(BF16).reshape(T * B, E).contiguous(), idx.reshape(-1).contiguous(), B * T)      # [B*T, E]
    R = B * T
    wqkv = torch.cat([par["self_attn.q_proj.weight"], par["self_attn.k_proj.weight"], par["self_attn.v_proj.weight"]], 0)
    bqkv = torch.cat([par["self_attn.q_proj.bias"], par["self_attn.k_proj.bias"], par["self_attn.v_proj.bias"]], 0)
    d = LayerDesc()
    d.B, d.N, d.E, d.F, d.H, d.Tp, d.m, d.r, d.num_cu = B, T, E, F, H, T, T, 0, 256
    pre_ln = bool(getattr(layer, "layer_norm_first", False))
    d.post_ln = int(not pre_ln)
    kp = padding_mask.to(torch.uint8).contiguous() if padding_mask is not None else None
    d.kpad = kp.data_ptr() if kp is not None else None
    d.wqkv, d.bqkv = wqkv.data_ptr(), bqkv.data_ptr()
    ident_g, ident_b = torch.ones(E, dtype=BF16, device=dev), torch.zeros(E, dtype=BF16, device=dev)
    if pre_ln:      # ln1 = final_layer_norm; the "next norm" does not exist for a lone layer: identity statistics are discarded
        n_in, _, _, _ = ops.ln_fwd(xb, par["self_attn_layer_norm.weight"], par["self_attn_layer_norm.bias"])
        la, lb = (par["final_layer_norm.weight"], par["final_layer_norm.bias"]), (ident_g, ident_b)
        d.x_in, d.stream_in = n_in.data_ptr(), xb.data_ptr()
    else:
        la = (par["self_attn_layer_norm.weight"], par["self_attn_layer_norm.bias"])
        lb = (par["final_layer_norm.weight"], par["final_layer_norm.bias"])
        d.x_in = xb.data_ptr()
    for f_, t_ in (("wo", par["self_attn.out_proj.weight"]), ("bo", par["self_attn.out_proj.bias"]), ("ln1_g", la[0]),
                   ("ln1_b", la[1]), ("w1", par["fc1.weight"]), ("b1", par["fc1.bias"]), ("w2", par["fc2.weight"]),
                   ("b2", par["fc2.bias"]), ("ln2_g", lb[0]), ("ln2_b", lb[1])):
        setattr(d, f_, t_.data_ptr())
    bufs = {}
    for f_, cols in (("qkv", 3 * E), ("ctx", E), ("s1", E), ("x1", E), ("hpre", F), ("h", F), ("s2", E), ("x_out", E), ("tmp", E)):
        bufs[f_] = torch.empty(R, cols, dtype=BF16, device=dev)
        setattr(d, f_, bufs[f_].data_ptr())
    f32 = torch.empty(B * H * T + 4 * R, dtype=torch.float32, device=dev)
    base = f32.data_ptr()
    d.lse = base
    d.mean1, d.rstd1 = base + 4 * (B * H * T), base + 4 * (B * H * T + R)
    d.mean2, d.rstd2 = base + 4 * (B * H * T + 2 * R), base + 4 * (B * H * T + 3 * R)
    _lib.call("w2vs_layer_fwd", C.byref(d), ops._stream())
    out = bufs["s2"] if pre_ln else bufs["x_out"]         # pre-LN: the layer's output is the stream itself
    res = torch.empty(T * B, E, dtype=BF16, device=dev)
    ops.gather_rows(out, idx.reshape(-1).contiguous(), B * T, scatter=True, out=res)
    return res.view(T, B, E).to(dt)


# The last encoder layer of a pre-training step computes only what the loss reads (masked frames); tests switch it off
# to compare the full encoder output.
SELECT_LAST_LAYER = True
# W2VS_ATTN_KEEP_BITS=1: park the attention-dropout decisions as bits in the forward, read them in the backward
KEEP_MASK_STORE = os.environ.get("W2VS_ATTN_KEEP_BITS", "0") == "1"
# the four weight gradients of a post-LN layer as one grouped launch (w2vs_gemm_tn_group); W2VS_GROUP_WGRADS=0 disables
PAIR_WGRADS = os.environ.get("W2VS_PAIR_WGRADS", "1") != "0"     # A/B: 0 = every layer launches its own weight gradients
GROUP_WGRADS = os.environ.get("W2VS_GROUP_WGRADS", "1") != "0"
# the selected-rows (last) layer's weight gradients join the grouped launch of its neighbour; W2VS_SEL_DEFER=0: four launches of their own (A/B)
SEL_DEFER = os.environ.get("W2VS_SEL_DEFER", "1") != "0"
LN_DEFER = os.environ.get("W2VS_LN_DEFER", "1") != "0"      # the layers' dgamma / dbeta partial sums wait for the grouped weight-gradient call
# how deferred weight gradients are put into grouped launches: "1" = packed by tile count (_WgradPacker), "0" = whole layer pairs
# (round 3), "auto" = packed only when a layer pair does not fit one round of the chip anyway (the large model: 192 tiles per
# layer, 192 + 64 = 256; 18.84 -> 18.44 ms per step).  The base model's pairs (216 tiles) stay: packed to 252 its launches take
# 175 us instead of 154 (the launch time follows the tile count there) and the 36-tile remainder costs what the saved launch gave
PACK_WGRADS = os.environ.get("W2VS_PACK_WGRADS", "auto")


class _WgradPacker:
    """Decides which deferred weight gradients go into which grouped launch (w2vs_layer_wgrads_parts).  An item is one GEMM of
    one layer (fc2, fc1, out_proj, fused QKV - the order they become available in) with its count of 256 x 256 output tiles;
    a launch takes items oldest first while they fit `cap` tiles (the CU count), at most 12 GEMMs of at most 4 layers.
    pack = False restores round 3's rule: two whole layers per launch."""
    BITS = (("fc2", 2), ("fc1", 1), ("out", 8), ("qkv", 4))

    @staticmethod
    def tile_counts(E, F):
        t = lambda n, k: ((n + 255) // 256) * ((k + 255) // 256)      # noqa: E731
        return {"fc2": t(E, F), "fc1": t(F, E), "out": t(E, E), "qkv": t(3 * E, E)}

    def __init__(self, stream, E, F, cap, pack):
        self.tiles = self.tile_counts(E, F)
        self.stream, self.cap, self.pack = stream, cap, pack
        self.items = []            # [desc, jj, bit, tiles]
        self.ln_due = {}           # jj -> True while the layer's LayerNorm partial sums are still to be reduced
        self.layers = {}           # jj -> desc of every layer that has something pending

    def add(self, d, jj):
        for name, bit in self.BITS:
            self.items.append([d, jj, bit, self.tiles[name]])
        self.ln_due[jj] = True
        self.layers[jj] = d

    def holds(self, jj):
        return jj in self.layers

    def pending_tiles(self):
        return sum(it[3] for it in self.items)

    def due(self):
        if not self.pack:
            return len(self.layers) >= 2
        return self.pending_tiles() >= self.cap

    def complete_prefix(self, ran):
        """how many of the `ran` layers run so far (backward order) have nothing pending, counted from the first"""
        return min(self.layers) if self.layers else ran

    def launch(self):
        chosen, tiles, jjs = [], 0, []
        for it in self.items:
            new_layer = it[1] not in jjs
            if len(chosen) == 12 or (new_layer and len(jjs) == 4):
                break
            if self.pack and chosen and tiles + it[3] > self.cap:
                continue
            chosen.append(it)
            tiles += it[3]
            if new_layer:
                jjs.append(it[1])
        if not chosen:
            return
        parts = {jj: 0 for jj in jjs}
        for it in chosen:
            parts[it[1]] |= it[2]
        for jj in jjs:
            if self.ln_due.pop(jj, False):
                parts[jj] |= 16
        ids = {id(it) for it in chosen}
        self.items = [it for it in self.items if id(it) not in ids]
        left = {it[1] for it in self.items}
        descs = [self.layers[jj] for jj in jjs]
        for jj in jjs:
            if jj not in left:
                del self.layers[jj]
        arr = (type(descs[0]) * len(descs))(*descs)
        pm = (C.c_int32 * len(descs))(*[parts[jj] for jj in jjs])
        _lib.call("w2vs_layer_wgrads_parts", arr, pm, len(descs), self.stream)



_POS_TABLES = {}


def _pos_table(E, dev):
    key = (E, str(dev))
    if key not in _POS_TABLES:
        _POS_TABLES[key] = host_rng.sinusoidal_table(8000 + 1 + 1, E, 1).to(dev).contiguous()  # wav2vec_S.py:341-347
    return _POS_TABLES[key]


def _qkv_pack(W, pre):
    q, k, v = W[pre + "self_attn.q_proj.weight"], W[pre + "self_attn.k_proj.weight"], W[pre + "self_attn.v_proj.weight"]
    qb, kb, vb = W[pre + "self_attn.q_proj.bias"], W[pre + "self_attn.k_proj.bias"], W[pre + "self_attn.v_proj.bias"]
    esz = q.element_size()
    if (k.data_ptr() == q.data_ptr() + q.numel() * esz and v.data_ptr() == k.data_ptr() + k.numel() * esz
            and kb.data_ptr() == qb.data_ptr() + qb.numel() * esz and vb.data_ptr() == kb.data_ptr() + kb.numel() * esz
            and q.storage_offset() + 3 * q.numel() <= q.untyped_storage().nbytes() // esz):
        # flat parameter storage keeps q, k, v adjacent: the fused [3E, E] weight is just a view
        E_ = q.shape[0]
        wv = torch.as_strided(q, (3 * E_, q.shape[1]), (q.shape[1], 1))
        bv = torch.as_strided(qb, (3 * E_,), (1,))
        return wv, bv
    w = torch.cat([W[pre + "self_attn.q_proj.weight"], W[pre + "self_attn.k_proj.weight"], W[pre + "self_attn.v_proj.weight"]], 0)
    b = torch.cat([W[pre + "self_attn.q_proj.bias"], W[pre + "self_attn.k_proj.bias"], W[pre + "self_attn.v_proj.bias"]], 0)
    return w, b


# ==================================================================================================
def grad_shapes(cfg, W) -> Dict[str, tuple]:
    """Arena layout = FORWARD order of the network (extractor, feature LN, projection, mask_emb, encoder
    LN, layers 0..L-1, quantizer, project_q, final_proj).  The backward pass therefore finalises the
    arena from its END towards its start, and the data-parallel exchange can all-reduce a growing
    suffix while earlier layers are still being differentiated.  q/k/v weights (and biases) of a layer
    are adjacent so the fused QKV GEMM writes one [3E, E] block; conv weights are tap-major."""
    def rank(n):
        if n.startswith("feature_extractor."):
            return (0, int(n.split(".")[2]))
        if n.startswith("layer_norm."):
            return (1, 0)
        if n.startswith("post_extract_proj."):
            return (2, 0)
        if n == "mask_emb":
            return (3, 0)
        if n.startswith("encoder.layer_norm."):
            return (4, 0)
        if n.startswith("encoder.layers."):
            return (5, int(n.split(".")[2]))
        if n.startswith("quantizer."):
            return (6, 0)
        if n.startswith("project_q."):
            return (7, 0)
        if n.startswith("final_proj."):
            return (8, 0)
        return (9, 0)

    def within(n):
        if n.startswith("encoder.layers."):
            tail = n.split(".", 3)[3]
            order = ["self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                     "self_attn.q_proj.bias", "self_attn.k_proj.bias", "self_attn.v_proj.bias"]
            return order.index(tail) if tail in order else len(order)
        return 0

    names = sorted(W.keys(), key=lambda n: (rank(n), within(n)))   # stable: keeps module order otherwise
    shapes = {}
    for n in names:
        shp = tuple(W[n].shape)
        if n.startswith("feature_extractor.conv_layers.") and n.endswith(".0.weight"):
            shp = (shp[0], shp[2], shp[1])
        shapes[n] = shp
    return shapes


def milestone_offset(A: "Arena", prefix: str) -> int:
    """Lowest arena offset among parameters whose name starts with ``prefix`` (= start of that group)."""
    offs = [o for n, (o, _, _) in A.offsets.items() if n.startswith(prefix)]
    return min(offs) if offs else A.numel


def _linear_bwd(dy, x, w_name, b_name, W, A, *, need_dx=True, dgelu_aux=None, add_aux=None):
    ops.linear_wgrad(dy, x, A.view(w_name), db_f32=A.view(b_name))
    if not need_dx:
        return None
    wt = ops.transpose2d(W[w_name])
    return ops.linear_dgrad(dy, wt, dgelu_aux=dgelu_aux, add_aux=add_aux)


def wgrad_overwrite_ranges(st: State, A: Arena):
    """Arena ranges (offset, numel) that ``backward(..., overwrite_wgrads=True)`` will WRITE rather than add to: the four weight
    matrices of every encoder layer whose weight gradients are deferred into the grouped single-writer launches
    (w2vs_layer_wgrads).  A trainer zeroes the rest of the arena only: the fill of these 85 M of the base model's 90 M
    gradients (340 MB) and their read inside the weight-gradient kernels disappear from the step.  Empty when the step does
    not defer (one kept layer, PAIR / GROUP switches off).  The last layer in selected-rows mode is deferred too (ws_s0 / ws_s1)."""
    if not (PAIR_WGRADS and GROUP_WGRADS and len(st.layers) > 1):
        return []
    E, F = st.cfg.encoder_embed_dim, st.cfg.encoder_ffn_embed_dim
    out = []
    for rec in st.layers:
        if rec["desc"].sel_idx and not SEL_DEFER:
            continue
        pre = f"encoder.layers.{rec['li']}."
        out.append((A.offsets[pre + "self_attn.q_proj.weight"][0], 3 * E * E))       # q | k | v weights are adjacent
        for n in ("self_attn.out_proj.weight", "fc1.weight", "fc2.weight"):
            off, numel, _ = A.offsets[pre + n]
            out.append((off, numel))
    return out


def backward(st: State, A: Arena, *, d_logits=None, d_pen=None, d_prob_ppl=None, d_out=None, on_ready=None,
             overwrite_wgrads: bool = False, wt_cache=None):
    """Accumulates parameter gradients into the arena.  d_logits fp32 [B*M, K+1] (rows (b, m));
    d_pen = dLoss/d features_pen, d_prob_ppl = dLoss/d prob_perplexity as fp32 DEVICE scalars (read by
    the kernels, never by the host: no sync); d_out bf16 [B, T, E] for features_only.
    on_ready(offset): called at milestones - every arena element at index >= offset is final.
    overwrite_wgrads: the ranges of ``wgrad_overwrite_ranges`` are written, not accumulated (the caller did not zero them).
    wt_cache: {"have": set, "store": dict} kept by the caller while the weights do not change (the micro-batches of one update,
    trainer.TrainStep with update_freq > 1): the transposed weights of the input-gradient GEMMs are made once per update, not
    once per backward.  None: made here, every time."""
    ready = on_ready if on_ready is not None else (lambda off: None)
    cfg, W = st.cfg, st.W
    B, T, C0, N, Tp = st.B, st.T, st.C0, st.N, st.Tp
    E = cfg.encoder_embed_dim
    H = cfg.encoder_attention_heads
    dev = st.feats.device
    p_in, p_feat, p_enc, p_att = st.p
    seed = st.seed
    R = B * N
    d_enc = None if getattr(st, "enc_is_sel", False) else ops.zeros((R, E), BF16, dev)
    d_feats_unmasked = None

    if st.features_only:
        ops.gather_rows(d_out.reshape(B * T, E).contiguous(), st.out_idx, B * T, scatter=True, out=d_enc)
    else:
        M, K = st.M, st.K
        RM = B * M
        dxf, dyq = ops.nce_bwd(d_logits, st.logits, st.nce_norms, st.xf, st.yq, st.neg, B, M, K, cfg.logit_temp)
        # final_proj
        dxm = _linear_bwd(dxf, st.xm, "final_proj.weight", "final_proj.bias", W, A)
        if getattr(st, "enc_is_sel", False):
            d_enc = dxm                    # the last layer ran on exactly these rows
        else:
            ops.gather_rows(dxm, st.token_idx, RM, scatter=True, out=d_enc)
        # project_q
        dq = _linear_bwd(dyq, st.q, "project_q.weight", "project_q.bias", W, A)
        # quantizer
        G, V = cfg.latent_groups, cfg.latent_vars
        vars2d = W["quantizer.vars"].view(G * V, -1)
        dql = ops.quant_bwd(dq, st.q_logits, vars2d, st.qst, G, V, st.tau, st.training,
                            1.0 if d_prob_ppl is not None else 0.0,
                            A.view("quantizer.vars").view(G * V, -1), noise=st.noise, seed=seed(4),
                            ppl_grad_dev=d_prob_ppl, bias=W["quantizer.weight_proj.bias"])
        d_yin = _linear_bwd(dql, st.y_in, "quantizer.weight_proj.weight", "quantizer.weight_proj.bias", W, A)
        if p_feat > 0:
            d_yin = ops.dropout(d_yin, p_feat, seed(3))
        d_feats_unmasked = ops.zeros((B * T, C0), BF16, dev)
        ops.gather_rows(d_yin, st.frame_idx, RM, scatter=True, out=d_feats_unmasked)

    ready(milestone_offset(A, "quantizer.") if "quantizer.weight_proj.weight" in A else A.numel)   # heads done
    # ------------------------------------------------------------------ encoder layers, reversed
    post_ln = not cfg.layer_norm_first
    F = cfg.encoder_ffn_embed_dim
    dx = d_enc
    d_stream = None                        # pre-LN: gradient of the residual stream (None behind the last layer)
    if st.layers:
        ws = ops.empty((R * (3 * E + F + 3 * E + E) + max(3 * E * E, E * F),), BF16, dev)
        delta = ops.empty((B * H * N,), torch.float32, dev)
        stream = ops._stream()
        wp = ws.data_ptr()
        offs = {}
        o = 0
        for f_, n_ in (("ws_e0", R * E), ("ws_e1", R * E), ("ws_e2", R * E), ("ws_f", R * F), ("ws_qkv", R * 3 * E),
                       ("d_in_a", R * E), ("wt_scratch", max(3 * E * E, E * F))):
            offs[f_] = wp + 2 * o
            o += n_
        d_in_bufs = [offs["d_in_a"], None]
        e3 = ops.empty((R, E), BF16, dev) if GROUP_WGRADS else None    # fourth [R,E] scratch: grouped weight gradients
        # Weight gradients of TWO layers in one launch (w2vs_layer_wgrads): a base layer's four are 108 tiles of 256^2 and need a
        # two-way K split with an exchange between workgroup pairs to fill the chip; two layers are 216 tiles with full-length
        # K loops and no exchange.  The first layer of a pair keeps its dY operands (ws_f, ws_e0, ws_qkv, ws_e3) while the second
        # one runs its backward on a second set of those four buffers.
        pair = PAIR_WGRADS and e3 is not None and len(st.layers) > 1
        # Round 4: the launches are PACKED.  A grouped launch is one round of the chip whatever its tile count, and a base layer is
        # 36 + 36 + 9 + 27 tiles: whole GEMMs of up to four layers are put together until they make ~252 of 256 tiles (a layer pair
        # is 216); the large model's 64 + 64 + 16 + 48 make exactly 256.  Operand sets rotate over NSET layers accordingly.
        tiles_per_layer = sum(_WgradPacker.tile_counts(E, F).values())
        ncu_ = st.layers[0]["desc"].num_cu or 256
        pack = pair and (PACK_WGRADS == "1" or (PACK_WGRADS == "auto" and 2 * tiles_per_layer > ncu_))
        NSET = 4 if pack else 2
        base_set = {"ws_e0": offs["ws_e0"], "ws_f": offs["ws_f"], "ws_qkv": offs["ws_qkv"], "ws_e3": e3.data_ptr() if e3 is not None else None}
        op_sets, more_sets = [base_set], []
        if pair:
            for _ in range(NSET - 1):
                ws2 = ops.empty((R * (E + F + 3 * E + E),), BF16, dev)
                p2 = ws2.data_ptr()
                more_sets.append(ws2)
                op_sets.append({"ws_e0": p2, "ws_f": p2 + 2 * R * E, "ws_qkv": p2 + 2 * R * (E + F), "ws_e3": p2 + 2 * R * (E + F + 3 * E)})
        alt = ops.empty((R, E), BF16, dev)
        d_in_bufs[1] = alt.data_ptr()
        # the selected-rows (last) layer joins the grouped weight-gradient launches when it gets its own scatter targets
        sel_scatter = ops.empty((2, R, E), BF16, dev) if (pair and SEL_DEFER and st.layers[-1]["desc"].sel_idx) else None
        ds_bufs = [ops.empty((R, E), BF16, dev), ops.empty((R, E), BF16, dev)] if not post_ln else None
        # one slab of LayerNorm partial sums per operand set: [2 norms][<= 768 blocks][2E] fp32 (w2vs_layer_desc.ln_part)
        ln_parts = ops.empty((NSET, 2 * 768 * 2 * E), torch.float32, dev) if (pair and LN_DEFER) else None
        cur = dx
        # every layer's four weights are transposed for the dgrad GEMMs in ONE launch (was 4 launches per layer)
        per_t = 3 * E * E + E * E + 2 * E * F
        have = wt_cache["have"] if wt_cache is not None else None
        wt_all = ops.empty((len(st.layers) * per_t,), BF16, dev) if wt_cache is None else None
        items = []
        for j_, rec in enumerate(st.layers):
            d = rec["desc"]
            if wt_cache is None:
                base, fresh = wt_all.data_ptr() + 2 * j_ * per_t, True
            else:                                          # a persistent buffer per layer (not the step arena), filled once per update
                buf_ = wt_cache["store"].get(rec["li"])
                if buf_ is None or buf_.numel() != per_t or buf_.device != torch.device(dev):
                    buf_ = wt_cache["store"][rec["li"]] = torch.empty((per_t,), dtype=BF16, device=dev)
                    have.discard(rec["li"])
                base, fresh = buf_.data_ptr(), rec["li"] not in have
                have.add(rec["li"])
            d.wqkv_t, d.wo_t = base, base + 2 * 3 * E * E
            d.w1_t, d.w2_t = base + 2 * 4 * E * E, base + 2 * (4 * E * E + E * F)
            if fresh:
                items += [(d.wqkv, d.wqkv_t, 3 * E, E), (d.wo, d.wo_t, E, E), (d.w1, d.w1_t, F, E), (d.w2, d.w2_t, E, F)]
        st._dgrad_w = {}
        conv_fresh = have is None or "conv" not in have
        for ci_ in range(1, len(cfg.conv_layers)):        # conv dgrad operands ride in the same launch
            _, ck, cs = cfg.conv_layers[ci_]
            buf, its = ops.conv_dgrad_weight_items((id(A), ci_), st.packed[ci_], ck, cs)
            st._dgrad_w[ci_] = buf
            if conv_fresh:
                items += its
        if have is not None:
            have.add("conv")
        if items:
            ops.transpose_multi(items)
        st._wt_all = wt_all
        tn_ws = ops.tn_workspace(dev)
        for rec in st.layers:
            rec["desc"].tn_ws, rec["desc"].tn_ws_bytes = tn_ws.data_ptr(), tn_ws.numel() * 4
        packer = _WgradPacker(stream, E, F, cap=ncu_, pack=pack)
        nl_ = len(st.layers)
        reported = 0                                       # layers (in backward order) whose milestone went out

        def report():
            # a milestone is reported only once every gradient at or above it is final, i.e. never while any of a layer's weight
            # gradients (or LayerNorm partial sums) are pending.  pre-LN: a layer's call also finalises the NEXT norm's gradient
            # (layer li+1's self_attn_layer_norm), so the arena is final from that layer's start only one layer later
            nonlocal reported
            done = packer.complete_prefix(ran)
            while reported < done:
                reported += 1
                if post_ln:
                    ready(milestone_offset(A, f"encoder.layers.{st.layers[nl_ - reported]['li']}."))
                elif reported > 1:
                    ready(milestone_offset(A, f"encoder.layers.{st.layers[nl_ - reported + 1]['li']}."))

        ran = 0
        for jj, rec in enumerate(reversed(st.layers)):
            li = rec["li"]
            pre = f"encoder.layers.{li}."
            d = rec["desc"]
            d.d_out = cur.data_ptr()
            tgt = d_in_bufs[jj & 1]
            d.d_in = tgt
            if not post_ln:
                d.d_stream_out = d_stream.data_ptr() if d_stream is not None else None
                d.d_stream_in = ds_bufs[jj & 1].data_ptr()
            for f_ in ("ws_e0", "ws_e1", "ws_e2", "ws_f", "ws_qkv", "wt_scratch"):
                setattr(d, f_, offs[f_])
            d.delta = delta.data_ptr()
            d.ws_e3 = e3.data_ptr() if e3 is not None else None
            deferred = bool(pair and (SEL_DEFER or not d.sel_idx))
            if d.sel_idx and deferred:
                d.ws_s0, d.ws_s1 = sel_scatter[0].data_ptr(), sel_scatter[1].data_ptr()
            d.defer_wgrads = 1 if deferred else 0
            d.wgrad_overwrite = 1 if (deferred and overwrite_wgrads) else 0
            d.ln_part, d.ln_part_bytes = None, 0
            if deferred:                                   # consecutive layers rotate over the operand sets
                while packer.holds(jj - NSET):             # ... whose previous user must have launched everything
                    packer.launch()
                    report()
                for f_, v_ in op_sets[jj % NSET].items():
                    setattr(d, f_, v_)
                if ln_parts is not None:
                    d.ln_part, d.ln_part_bytes = ln_parts[jj % NSET].data_ptr(), ln_parts[jj % NSET].numel() * 4
            off_w = A.offsets[pre + "self_attn.q_proj.weight"][0]
            off_b = A.offsets[pre + "self_attn.q_proj.bias"][0]
            fp = A.flat.data_ptr()
            d.g_wqkv, d.g_bqkv = fp + 4 * off_w, fp + 4 * off_b
            na, nb = rec["norm_a"], rec["norm_b"]
            for f_, n_ in (("g_wo", pre + "self_attn.out_proj.weight"), ("g_bo", pre + "self_attn.out_proj.bias"),
                           ("g_ln1_g", na[0]), ("g_ln1_b", na[1]), ("g_w1", pre + "fc1.weight"), ("g_b1", pre + "fc1.bias"),
                           ("g_w2", pre + "fc2.weight"), ("g_b2", pre + "fc2.bias"), ("g_ln2_g", nb[0]), ("g_ln2_b", nb[1])):
                setattr(d, f_, fp + 4 * A.offsets[n_][0])
            _lib.call("w2vs_layer_bwd", C.byref(d), stream)
            ran = jj + 1
            if deferred:
                packer.add(d, jj)
                while packer.due():
                    packer.launch()
            report()
            if jj & 1:
                cur = alt
            else:
                cur = ws[(3 * R * E + R * F + 3 * R * E):(3 * R * E + R * F + 3 * R * E) + R * E].view(R, E)
            if not post_ln:
                d_stream = ds_bufs[jj & 1]
        while packer.items:                                 # what is left goes out in as few launches as it takes
            packer.launch()
        report()
        dx = cur
        st._bwd_ws = (ws, alt, delta, e3, ds_bufs, more_sets, sel_scatter, ln_parts)
    if post_ln:
        d_x0 = dx
    elif st.layers:
        pre = f"encoder.layers.{st.layers[0]['li']}."
        d_x0, _ = ops.ln_bwd(st.x0, W[pre + "self_attn_layer_norm.weight"], W[pre + "self_attn_layer_norm.bias"],
                             st.n0_mean, st.n0_rstd, A.view(pre + "self_attn_layer_norm.weight"),
                             A.view(pre + "self_attn_layer_norm.bias"), dy=dx, dsum=d_stream)
    else:
        d_x0, _ = ops.ln_bwd(st.x0, W["encoder.layer_norm.weight"], W["encoder.layer_norm.bias"], st.fin_mean,
                             st.fin_rstd, A.view("encoder.layer_norm.weight"), A.view("encoder.layer_norm.bias"), dy=d_enc)

    ready(milestone_offset(A, "encoder.layers.0."))
    # ------------------------------------------------------------------ prologue
    table = _pos_table(E, dev)
    d_xproj = ops.enc_prologue_bwd(
        d_x0.view(B, N, E), st.xproj, st.mask_dev, st.pad_dev, st.pos, W["mask_emb"], table,
        W["encoder.layer_norm.weight"], W["encoder.layer_norm.bias"], st.p_mean, st.p_rstd, st.src, st.copy_start,
        st.copy_list, Tp, A.view("mask_emb"), A.view("encoder.layer_norm.weight"), A.view("encoder.layer_norm.bias"),
        apply_ln=post_ln, p_in=p_in, seed_in=seed(1), p_enc=p_enc, seed_enc=seed(2))
    if cfg.feature_grad_mult <= 0:
        # extractor ran under no_grad in the reference (wav2vec2.py:550-552); the projection and
        # the feature LayerNorm still train
        pass
    if "post_extract_proj.weight" in W:
        d_feats = _linear_bwd(d_xproj.view(B * T, E), st.feats.view(B * T, C0), "post_extract_proj.weight",
                              "post_extract_proj.bias", W, A, add_aux=d_feats_unmasked)
    else:
        d_feats = d_xproj.view(B * T, C0)
        if d_feats_unmasked is not None:
            raise W2vsError("conv dim == encoder dim with the pre-training head is not built")
    ready(milestone_offset(A, "post_extract_proj.") if "post_extract_proj.weight" in A else milestone_offset(A, "mask_emb"))
    # ------------------------------------------------------------------ feature LN + penalty + GradMultiply
    convs = cfg.conv_layers
    last = st.conv[-1]
    numel = float(B * T * C0)
    gm = cfg.feature_grad_mult
    if gm <= 0:
        ops.ln_bwd(st.y_last, W["layer_norm.weight"], W["layer_norm.bias"], st.f_mean, st.f_rstd,
                   A.view("layer_norm.weight"), A.view("layer_norm.bias"), dy=d_feats, want_dx=False)
        ready(0)
        return
    aux = last.get("pre") if (len(st.conv) > 1 and not last["ln"]) else None
    d_cur, _ = ops.ln_bwd(st.y_last, W["layer_norm.weight"], W["layer_norm.bias"], st.f_mean, st.f_rstd,
                          A.view("layer_norm.weight"), A.view("layer_norm.bias"), dy=d_feats, aux=aux, out_scale=gm,
                          pen_coef=(1.0 / numel) if d_pen is not None else 0.0, pen_coef_dev=d_pen)
    # d_cur: grad wrt conv_i pre-activation (plain layers) or wrt conv_i's post-GELU output (LN layers / layer 0)
    for i in range(len(convs) - 1, 0, -1):
        rec = st.conv[i]
        dim, k, s = convs[i]
        if rec["ln"]:
            d_c, _ = ops.ln_bwd(rec["c"], W[_pname(i, "2.1.weight")], W[_pname(i, "2.1.bias")], rec["mean"], rec["rstd"],
                                A.view(_pname(i, "2.1.weight")), A.view(_pname(i, "2.1.bias")), dy=d_cur, gelu=True)
        else:
            d_c = d_cur
        x_in = rec["x_in"]
        ops.conv_cl_wgrad(d_c, x_in, k, s, A.view(_pname(i, "0.weight")).view(dim, -1),
                          db_f32=A.view(_pname(i, "0.bias")) if _pname(i, "0.bias") in A else None)
        prev = st.conv[i - 1]
        prev_aux = prev.get("pre") if (i - 1 >= 1 and not prev["ln"]) else None
        d_cur = ops.conv_cl_dgrad(d_c, st.packed[i], k, s, x_in.shape[1], mul_aux=prev_aux,
                                  wprep=getattr(st, "_dgrad_w", {}).get(i))
    dim0, k0, s0 = convs[0]
    r0 = st.conv[0]
    if cfg.extractor_mode == "default":
        ops.conv0_gn_bwd(st.source, st.packed[0], W[_pname(0, "2.weight")], W[_pname(0, "2.bias")], r0["gstat"], d_cur,
                         k0, s0, A.view(_pname(0, "0.weight")).view(dim0, k0), A.view(_pname(0, "2.weight")),
                         A.view(_pname(0, "2.bias")), conv_bias=W.get(_pname(0, "0.bias")),
                         dconv_bias=A.view(_pname(0, "0.bias")) if _pname(0, "0.bias") in A else None)
        ready(0)
        return
    ops.conv0_bwd(st.source, st.packed[0], W[_pname(0, "2.1.weight")], W[_pname(0, "2.1.bias")], r0["mean"],
                  r0["rstd"], d_cur, k0, s0, A.view(_pname(0, "0.weight")).view(dim0, k0), A.view(_pname(0, "2.1.weight")),
                  A.view(_pname(0, "2.1.bias")), conv_bias=W.get(_pname(0, "0.bias")),
                  dconv_bias=A.view(_pname(0, "0.bias")) if _pname(0, "0.bias") in A else None)


def _attn_block_bwd(st, rec, pre, d_a, x_in, add_to_dx, A):
    """out_proj -> attention -> fused QKV projection, backward.  Returns grad wrt the block input."""
    cfg, W = st.cfg, st.W
    B, N, Tp = st.B, st.N, st.Tp
    E, H = cfg.encoder_embed_dim, cfg.encoder_attention_heads
    R = B * N
    li = rec["li"]
    d_ctx = _linear_bwd(d_a, rec["ctx"].view(R, E), pre + "self_attn.out_proj.weight", pre + "self_attn.out_proj.bias", W, A)
    dqkv = ops.attn_bwd(d_ctx.view(B, N, E), rec["qkv"].view(B, N, 3 * E), rec["ctx"], rec["lse"], H, Tp, st.m, st.r,
                        kpad=st.kpad, p_drop=st.p[3], seed=st.seed(100 + 4 * li)).view(R, 3 * E)
    # fused [3E, E] weight block: q, k, v views are adjacent in the arena
    off_w = A.offsets[pre + "self_attn.q_proj.weight"][0]
    off_b = A.offsets[pre + "self_attn.q_proj.bias"][0]
    dW = A.flat[off_w:off_w + 3 * E * E].view(3 * E, E)
    db = A.flat[off_b:off_b + 3 * E]
    ops.linear_wgrad(dqkv, x_in, dW, db_f32=db)
    wqkv, _ = _qkv_pack(W, pre)
    wt = ops.transpose2d(wqkv)
    return ops.linear_dgrad(dqkv, wt, add_aux=add_to_dx)
