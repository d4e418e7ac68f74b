"""CPU oracle for the wav2vec-S pre-training hot path.  TEST INFRASTRUCTURE ONLY.

This is a plain fp32 PyTorch/numpy *restatement* of the reference algorithm
(biaofuxmu/wav2vec-S, vendored fairseq).  It exists to check the HIP path; it is
never the thing shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product package
``wav2vec-s_amd`` must never import anything from ``oracle/``.

Pinning: the reference holds NO test, golden vector or fixture for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, imported in the build container through ``oracle/ref_import.py``; the
resulting vectors are committed under ``tests/golden/`` by
``tests/golden/gen_golden.py`` and re-checked on every CPU test run
(``tests/test_oracle_golden.py``; ``tests/test_oracle_vs_reference.py`` re-runs
the live comparison whenever ``/root/reference`` is present).

Every function cites the reference lines it restates; paths are relative to
``/root/reference/fairseq/fairseq/`` ("fs/").  Parameter names are the reference's
``state_dict`` keys so a reference checkpoint drives the oracle unchanged.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------
# configuration (field names = fs/models/wav2vec/wav2vec_S.py:43-311 Wav2VecSConfig)
# ----------------------------------------------------------------------------------
@dataclass
class OracleCfg:
    extractor_mode: str = "layer_norm"
    encoder_layers: int = 12
    encoder_embed_dim: int = 768
    encoder_ffn_embed_dim: int = 3072
    encoder_attention_heads: int = 12
    layer_norm_first: bool = False
    conv_feature_layers: str = "[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] * 2"
    conv_bias: bool = False
    final_dim: int = 256
    latent_vars: int = 320
    latent_groups: int = 2
    latent_dim: int = 0
    logit_temp: float = 0.1
    num_negatives: int = 100
    mask_prob: float = 0.65
    mask_length: int = 10
    feature_grad_mult: float = 0.1
    required_seq_len_multiple: int = 2
    loss_weights: Tuple[float, ...] = (0.1, 10.0)

    @property
    def conv_layers(self) -> List[Tuple[int, int, int]]:
        return eval(self.conv_feature_layers)

    @property
    def layer_norm_num(self) -> int:
        # fs/models/wav2vec/wav2vec_S.py:325
        return 1 if self.encoder_layers == 12 else 7


# ----------------------------------------------------------------------------------
# host-side integer / RNG pieces (bit-exact contracts)
# ----------------------------------------------------------------------------------
def conv_out_lengths(L: int, conv_layers) -> List[int]:
    """fs/data/audio/raw_audio_dataset.py:194-202 (no padding, floor)."""
    out = []
    for _, k, s in conv_layers:
        L = (L - k) // s + 1
        out.append(L)
    return out


def compute_mask_indices(
    shape,
    padding_mask,
    mask_prob: float,
    mask_length: int,
    mask_type: str = "static",
    mask_other: float = 0.0,
    min_masks: int = 0,
) -> np.ndarray:
    """Span-mask sampler, fs/data/data_utils.py:389-513 (overlap-allowed branch only;
    the ``no_overlap`` branch uses the removed ``np.int`` and is unreachable with
    numpy>=1.24, data_utils.py:481).  Consumes the numpy GLOBAL RandomState in the
    reference's order: 1 rand() [+1 per row with a padding mask], one
    choice(replace=False) per row, one more per row that is longer than the minimum."""
    bsz, all_sz = shape
    mask = np.full((bsz, all_sz), False)
    all_num_mask = int(mask_prob * all_sz / float(mask_length) + np.random.rand())
    all_num_mask = max(min_masks, all_num_mask)
    rows = []
    for i in range(bsz):
        if padding_mask is not None:
            sz = all_sz - int(padding_mask[i].long().sum().item())
            num_mask = int(mask_prob * sz / float(mask_length) + np.random.rand())
            num_mask = max(min_masks, num_mask)
        else:
            sz = all_sz
            num_mask = all_num_mask
        if mask_type == "static":
            lengths = np.full(num_mask, mask_length)
        elif mask_type == "uniform":
            lengths = np.random.randint(mask_other, mask_length * 2 + 1, size=num_mask)
        elif mask_type == "normal":
            lengths = np.random.normal(mask_length, mask_other, size=num_mask)
            lengths = [max(1, int(round(x))) for x in lengths]
        elif mask_type == "poisson":
            lengths = np.random.poisson(mask_length, size=num_mask)
            lengths = [int(round(x)) for x in lengths]
        else:
            raise Exception("unknown mask selection " + mask_type)
        if sum(lengths) == 0:
            lengths[0] = min(mask_length, sz - 1)
        min_len = min(lengths)
        if sz - min_len <= num_mask:
            min_len = sz - num_mask - 1
        starts = np.random.choice(sz - min_len, num_mask, replace=False)
        idc = np.asarray(
            [starts[j] + off for j in range(len(starts)) for off in range(lengths[j])]
        )
        rows.append(np.unique(idc[idc < sz]))
    min_len = min(len(m) for m in rows)
    for i, idc in enumerate(rows):
        if len(idc) > min_len:
            idc = np.random.choice(idc, min_len, replace=False)
        mask[i, idc] = True
    return mask


def sample_negative_indices(bsz: int, num: int, n_negatives: int) -> torch.Tensor:
    """fs/models/wav2vec/wav2vec2.py:471-527, n_negatives>0 and cross_sample=0 branch.
    One torch.randint on the CPU default generator, "+1 where >= own position",
    then + b*num row offset.  Returns int64 (bsz, n_negatives*num) indices into the
    flattened (bsz*num) target rows."""
    assert num > 1, (bsz, num)
    tszs = torch.arange(num).unsqueeze(-1).expand(-1, n_negatives).flatten()
    neg = torch.randint(low=0, high=num - 1, size=(bsz, n_negatives * num))
    neg[neg >= tszs] += 1
    for i in range(1, bsz):
        neg[i] += i * num
    return neg


def sinusoidal_table(num_embeddings: int, dim: int, padding_idx: int = 1) -> torch.Tensor:
    """fs/modules/sinusoidal_positional_embedding.py:35-58: sin half then cos half,
    freq = exp(-i*log(1e4)/(half-1)), row ``padding_idx`` zeroed."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(num_embeddings, -1)
    if dim % 2 == 1:
        emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
    emb[padding_idx, :] = 0
    return emb


def positions_from_padding(pad: torch.Tensor, padding_idx: int = 1) -> torch.Tensor:
    """fs/utils.py:250-260 applied to the BOOL padding mask as wav2vec_S.py:361-367
    does: non-padded frame j gets padding_idx + (#non-padded frames up to j); padded
    frames get padding_idx (the zero row)."""
    nonpad = (~pad).int()
    return (torch.cumsum(nonpad, dim=1) * nonpad).long() + padding_idx


def block_structure(seq_len: int, main_context: int, right_context: int):
    """Integer structure of gen_block_attn_mask, fs/models/wav2vec/wav2vec_S.py:444-489.

    Returns (rc_idx int64 [R], rc_oob bool [R], masked bool [N, N]) with
    N = seq_len + R, R = (seq_len // m) * r.  masked[q, k] True == the reference adds
    -1e4 to that score."""
    m, r = main_context, right_context
    nb = seq_len // m
    block_idx = torch.arange(seq_len) // m
    if r == 0:
        return (
            torch.zeros(0, dtype=torch.long),
            torch.zeros(0, dtype=torch.bool),
            block_idx.unsqueeze(1) < block_idx.unsqueeze(0),
        )
    rc_block = torch.arange(nb).repeat_interleave(r)
    rc_idx = ((torch.arange(nb).unsqueeze(1) + 1) * m + torch.arange(r).unsqueeze(0)).reshape(-1)
    rc_oob = rc_idx > seq_len - 1
    rc_idx = rc_idx.clamp(0, seq_len - 1)
    full = torch.cat([block_idx, rc_block])
    m1 = full.unsqueeze(1) < block_idx.unsqueeze(0)
    m2 = full.unsqueeze(1) != rc_block.unsqueeze(0)
    return rc_idx, rc_oob, torch.cat([m1, m2], dim=1)


# ----------------------------------------------------------------------------------
# floating-point pieces
# ----------------------------------------------------------------------------------
def conv_feature_extractor(source: torch.Tensor, P: Dict[str, torch.Tensor], cfg: OracleCfg,
                           collect: Optional[dict] = None) -> torch.Tensor:
    """ConvFeatureExtractionModel.forward, fs/models/wav2vec/wav2vec2.py:702-781.
    Returns B x C x T (reference layout)."""
    x = source.unsqueeze(1)
    pre = "feature_extractor.conv_layers."
    for i, (dim, k, s) in enumerate(cfg.conv_layers):
        w = P[f"{pre}{i}.0.weight"]
        b = P.get(f"{pre}{i}.0.bias")
        x = F.conv1d(x, w, b, stride=s)
        if cfg.extractor_mode == "layer_norm" and i < cfg.layer_norm_num:
            # TransposeLast -> Fp32LayerNorm(dim) -> TransposeLast  (:733-743)
            x = F.layer_norm(x.transpose(1, 2).float(), (dim,), P[f"{pre}{i}.2.1.weight"].float(),
                             P[f"{pre}{i}.2.1.bias"].float(), 1e-5).transpose(1, 2)
        elif cfg.extractor_mode == "default" and i == 0:
            # Fp32GroupNorm(dim, dim)  (:744-750)
            x = F.group_norm(x.float(), dim, P[f"{pre}{i}.2.weight"].float(),
                             P[f"{pre}{i}.2.bias"].float(), 1e-5)
        x = F.gelu(x)
        if collect is not None:
            collect[f"conv{i}"] = x
    return x


def encoder_layer(x, P, pre, cfg: OracleCfg, add_mask, collect=None, drop=None):
    """TransformerSentenceEncoderLayer.forward, fs/models/wav2vec/wav2vec2.py:921-978,
    with MultiheadAttention's fast path (fs/modules/multihead_attention.py:161-193) spelt
    out: q scaled by head_dim**-0.5, additive mask, softmax, P.V, out_proj.
    x: N x B x C.  add_mask: B x 1 x N x N additive fp32 (attn mask + key padding).
    ``drop``: injected dropout decisions of a training-mode run, as MULTIPLICATIVE masks (0 or 1/(1-p)):
    "attn" B x H x N x N on the softmax output (``dropout_p`` of F.multi_head_attention_forward,
    multihead_attention.py:161-193), "drop1" N x B x C on the attention branch (dropout1, wav2vec2.py:945 / :966),
    "drop3" N x B x C on the FFN output (dropout3, :953 / :974); dropout2 is ``activation_dropout`` = 0 in
    the wav2vec-S yamls."""
    drop = drop or {}
    N, B, C = x.shape
    H = cfg.encoder_attention_heads
    D = C // H

    def lin(t, name):
        return F.linear(t, P[f"{pre}{name}.weight"], P[f"{pre}{name}.bias"])

    def attn(t):
        q = lin(t, "self_attn.q_proj").view(N, B, H, D).permute(1, 2, 0, 3) * (D ** -0.5)
        k = lin(t, "self_attn.k_proj").view(N, B, H, D).permute(1, 2, 0, 3)
        v = lin(t, "self_attn.v_proj").view(N, B, H, D).permute(1, 2, 0, 3)
        s = q @ k.transpose(-1, -2) + add_mask
        p = torch.softmax(s, dim=-1)
        if "attn" in drop:
            p = p * drop["attn"]
        o = (p @ v).permute(2, 0, 1, 3).reshape(N, B, C)
        if collect is not None:
            collect[pre + "attn_ctx"] = o
        return lin(o, "self_attn.out_proj")

    def ln(t, name):
        return F.layer_norm(t, (C,), P[f"{pre}{name}.weight"], P[f"{pre}{name}.bias"], 1e-5)

    d1, d3 = drop.get("drop1", 1.0), drop.get("drop3", 1.0)
    if cfg.layer_norm_first:
        x = x + attn(ln(x, "self_attn_layer_norm")) * d1
        h = F.gelu(lin(ln(x, "final_layer_norm"), "fc1").float())
        x = x + lin(h, "fc2") * d3
    else:
        x = ln(x + attn(x) * d1, "self_attn_layer_norm")
        h = F.gelu(lin(x, "fc1").float())
        x = ln(x + lin(h, "fc2") * d3, "final_layer_norm")
    return x


def blockwise_encoder(x, P, cfg: OracleCfg, main_context: int, right_context: int,
                      padding_mask=None, layer_keep: Optional[List[bool]] = None, collect=None, drop=None):
    """BlockwiseTransformerEncoder.extract_features + TransformerEncoder.forward,
    fs/models/wav2vec/wav2vec_S.py:355-440, wav2vec2.py:828-834.  x: B x T x C (masked,
    projected features).  Dropout is the identity unless its decisions are injected through ``drop``
    (multiplicative masks): "encoder" B x T' x C for F.dropout(x, p=self.dropout) of wav2vec_S.py:386 - applied
    BEFORE gen_block_attn_mask appends the right-context copies, which therefore carry the same decisions -
    and "layer{i}" -> the dict ``encoder_layer`` takes."""
    drop = drop or {}
    B, T, C = x.shape
    if padding_mask is not None:
        x = x.masked_fill(padding_mask.unsqueeze(-1), 0.0)
        pad = padding_mask
    else:
        pad = torch.zeros(B, T, dtype=torch.bool)
    table = sinusoidal_table(8000 + 1 + 1, C, 1)  # wav2vec_S.py:341-347
    pos = table.index_select(0, positions_from_padding(pad).view(-1)).view(B, T, C)
    x = x + pos
    if not cfg.layer_norm_first:
        x = F.layer_norm(x, (C,), P["encoder.layer_norm.weight"], P["encoder.layer_norm.bias"], 1e-5)
    mult = cfg.required_seq_len_multiple
    pad_len = (-T) % mult
    if pad_len:
        x = F.pad(x, (0, 0, 0, pad_len))
        # wav2vec_S.py:378-384: a fresh mask when none was given, else pad with True
        if padding_mask is None:
            pad = torch.zeros(B, T + pad_len, dtype=torch.bool)
            pad[:, -pad_len:] = True
        else:
            pad = F.pad(pad, (0, pad_len), value=True)
    Tp = T + pad_len
    if "encoder" in drop:
        x = x * drop["encoder"]
    if collect is not None:
        collect["enc_in"] = x
    x = x.transpose(0, 1)  # T' x B x C
    rc_idx, rc_oob, masked = block_structure(Tp, main_context, right_context)
    if right_context > 0:
        pad = torch.cat([pad, pad.index_select(1, rc_idx) | rc_oob.unsqueeze(0)], dim=1)
        x = torch.cat([x, x.index_select(0, rc_idx)], dim=0)
    add = torch.zeros(masked.shape).masked_fill(masked, -1e4)  # :486-487 (finite, not -inf)
    add = add.view(1, 1, *masked.shape) + torch.zeros(B, 1, 1, masked.shape[1]).masked_fill(
        pad.view(B, 1, 1, -1), float("-inf"))
    for i in range(cfg.encoder_layers):
        if layer_keep is None or layer_keep[i]:
            x = encoder_layer(x, P, f"encoder.layers.{i}.", cfg, add, collect, drop.get(f"layer{i}"))
            if collect is not None:
                collect[f"layer{i}"] = x
    x = x[:Tp].transpose(0, 1)
    if pad_len:
        x = x[:, :-pad_len]
    if cfg.layer_norm_first:
        x = F.layer_norm(x, (C,), P["encoder.layer_norm.weight"], P["encoder.layer_norm.bias"], 1e-5)
    return x


def gumbel_quantize(y, P, cfg: OracleCfg, tau: float, noise: Optional[torch.Tensor],
                    force_idx: Optional[torch.Tensor] = None):
    """GumbelVectorQuantizer.forward, fs/modules/gumbel_vector_quantizer.py:141-202.
    y: B x M x 512.  ``noise`` None = eval (hard one-hot of the argmax); otherwise the
    Gumbel sample g (shape (B*M*G, V)) that F.gumbel_softmax would have drawn:
    y_soft = softmax((logits+g)/tau); out = onehot(argmax y_soft) - y_soft.detach() + y_soft.
    ``force_idx`` (B*M, G) overrides the argmax SELECTION only (test aid: a bf16 implementation can
    flip a near-tied argmax; with the selection pinned the remaining fp math is comparable); the
    returned idx is still the natural fp32 argmax."""
    B, M, Fd = y.shape
    G, V = cfg.latent_groups, cfg.latent_vars
    logits = F.linear(y.reshape(-1, Fd), P["quantizer.weight_proj.weight"],
                      P["quantizer.weight_proj.bias"]).view(B * M * G, V)
    k = logits.argmax(-1)
    hard = torch.zeros_like(logits).scatter_(-1, k.view(-1, 1), 1.0).view(B * M, G, V)
    hp = hard.float().mean(0)
    code_ppl = torch.exp(-torch.sum(hp * torch.log(hp + 1e-7), dim=-1)).sum()
    ap = torch.softmax(logits.view(B * M, G, V).float(), dim=-1).mean(0)
    prob_ppl = torch.exp(-torch.sum(ap * torch.log(ap + 1e-7), dim=-1)).sum()
    if noise is not None:
        soft = torch.softmax((logits.float() + noise) / tau, dim=-1)
        idx = soft.argmax(-1)
        use = idx if force_idx is None else force_idx.reshape(-1).long()
        onehot = torch.zeros_like(soft).scatter_(-1, use.view(-1, 1), 1.0)
        sel = onehot - soft.detach() + soft
    else:
        idx = k
        use = idx if force_idx is None else force_idx.reshape(-1).long()
        sel = torch.zeros_like(logits).scatter_(-1, use.view(-1, 1), 1.0)
    vars_ = P["quantizer.vars"]  # 1 x (G*V) x D
    D = vars_.shape[-1]
    q = (sel.view(B * M, G * V, 1) * vars_).view(B * M, G, V, D).sum(-2).view(B, M, G * D)
    return q, idx.view(B * M, G), prob_ppl, code_ppl


def compute_logits(x, y, neg_idx, cfg: OracleCfg):
    """sample_negatives' gather + compute_preds + get_logits,
    fs/models/wav2vec/wav2vec2.py:521-542, 671-675.
    x, y: B x M x C; neg_idx: B x (K*M) int64.  Returns ((K+1) x B x M preds, (M*B) x (K+1) logits)."""
    B, M, C = y.shape
    K = neg_idx.shape[1] // M
    negs = y.reshape(-1, C)[neg_idx.view(-1)].view(B, M, K, C).permute(2, 0, 1, 3)
    neg_is_pos = (y == negs).all(-1)
    targets = torch.cat([y.unsqueeze(0), negs], dim=0)
    preds = torch.cosine_similarity(x.float(), targets.float(), dim=-1) / cfg.logit_temp
    if neg_is_pos.any():
        preds[1:][neg_is_pos] = float("-inf")
    logits = preds.transpose(0, 2).reshape(-1, K + 1)
    return preds, logits


def criterion(logits, prob_ppl, features_pen, num_vars: int, cfg: OracleCfg):
    """Wav2vecCriterion.forward (infonce), fs/criterions/wav2vec_criterion.py:64-157."""
    target = torch.zeros(logits.shape[0], dtype=torch.long)
    loss0 = F.cross_entropy(logits.float(), target, reduction="sum")
    sample_size = target.numel()
    extra = [(num_vars - prob_ppl) / num_vars, features_pen]
    losses = [loss0]
    loss = loss0
    for p, coef in zip(extra, cfg.loss_weights):
        if coef != 0 and p is not None:
            t = coef * p.float() * sample_size
            loss = loss + t
            losses.append(t)
    with torch.no_grad():
        mx = logits.argmax(-1) == 0
        mn = logits.argmin(-1) == 0
        correct = int(mx.long().sum().item() - (mx & mn).long().sum().item())
    return loss, sample_size, {"losses": losses, "correct": correct, "count": float(mx.numel())}


def forward_loss(P: Dict[str, torch.Tensor], source: torch.Tensor, cfg: OracleCfg, *,
                 mask_indices: torch.Tensor, neg_idx: torch.Tensor, main_context: int,
                 right_context: int, tau: float = 2.0, gumbel_noise: Optional[torch.Tensor] = None,
                 layer_keep: Optional[List[bool]] = None, collect: Optional[dict] = None,
                 force_code_idx: Optional[torch.Tensor] = None, drop: Optional[dict] = None):
    """Wav2Vec2Model.forward (fs/models/wav2vec/wav2vec2.py:544-658) + criterion, with
    every host-RNG draw INJECTED (mask_indices B x T bool, neg_idx, context sizes,
    gumbel noise, LayerDrop keeps) so that two implementations can be compared on identical
    draws.  Dropouts are off unless their decisions are injected too,
    through ``drop`` (multiplicative masks, 0 or 1/(1-p)): "input" B x T x C on the projected features
    (dropout_input, wav2vec2.py:570, before apply_mask), "features" B x M x C0 on the quantizer's input
    (dropout_features, :571 - the reference draws it for all B x T frames and then keeps the masked
    ones, :590-595; only those decisions matter), and the encoder's own ("encoder", "layer{i}":
    ``blockwise_encoder``)."""
    drop = drop or {}
    feats = conv_feature_extractor(source, P, cfg, collect)
    if cfg.feature_grad_mult != 1.0:
        # GradMultiply (fs/modules/grad_multiply.py:9-18): identity fwd, grad * scale
        s = cfg.feature_grad_mult
        feats = feats * s + (feats * (1.0 - s)).detach()
    features_pen = feats.float().pow(2).mean()
    feats = feats.transpose(1, 2)
    C0 = feats.shape[-1]
    feats = F.layer_norm(feats, (C0,), P["layer_norm.weight"], P["layer_norm.bias"], 1e-5)
    unmasked = feats.clone()
    if "post_extract_proj.weight" in P:  # None when conv dim == embed dim (wav2vec2.py:320-324)
        x = F.linear(feats, P["post_extract_proj.weight"], P["post_extract_proj.bias"])
    else:
        x = feats
    B, T, C = x.shape
    if "input" in drop:
        x = x * drop["input"]
    x = torch.where(mask_indices.unsqueeze(-1), P["mask_emb"].view(1, 1, C).expand(B, T, C), x)
    y = unmasked[mask_indices].view(B, -1, C0)
    if "features" in drop:
        y = y * drop["features"]
    if collect is not None:
        collect.update(features=feats, x_masked=x, y_in=y)
    x = blockwise_encoder(x, P, cfg, main_context, right_context, None, layer_keep, collect, drop)
    q, idx, prob_ppl, code_ppl = gumbel_quantize(y, P, cfg, tau, gumbel_noise, force_code_idx)
    yq = F.linear(q, P["project_q.weight"], P["project_q.bias"])
    xm = x[mask_indices].view(B, -1, C)
    xf = F.linear(xm, P["final_proj.weight"], P["final_proj.bias"])
    preds, logits = compute_logits(xf, yq, neg_idx, cfg)
    num_vars = cfg.latent_vars * cfg.latent_groups
    loss, sample_size, log = criterion(logits, prob_ppl, features_pen, num_vars, cfg)
    if collect is not None:
        collect.update(enc_out=x, q=q, q_idx=idx, yq=yq, xf=xf, preds=preds, logits=logits)
    out = dict(loss=loss, sample_size=sample_size, features_pen=features_pen,
               prob_perplexity=prob_ppl, code_perplexity=code_ppl, logits=logits,
               correct=log["correct"], count=log["count"], losses=log["losses"])
    return out


def polynomial_decay_lr(num_updates: int, lr: float, warmup_updates: int, total_num_update: float,
                        end_learning_rate: float = 0.0, power: float = 1.0) -> float:
    """PolynomialDecayLRSchedule.step_update, fs/optim/lr_scheduler/polynomial_decay_schedule.py:74-89: the rate applied
    to the update that follows ``num_updates`` completed ones (linear warm-up from 0, then polynomial decay)."""
    if warmup_updates > 0 and num_updates <= warmup_updates:
        return num_updates / float(warmup_updates) * lr
    if num_updates >= total_num_update:
        return end_learning_rate
    pct = 1 - (num_updates - warmup_updates) / (total_num_update - warmup_updates)
    return (lr - end_learning_rate) * pct ** power + end_learning_rate


def clip_coef(total_norm: float, max_norm: float) -> float:
    """fs/utils.py:379-383: (max_norm / (total_norm + 1e-6)).clamp_(max=1)."""
    return min(1.0, max_norm / (total_norm + 1e-6)) if max_norm > 0 else 1.0


def adam_update(p: torch.Tensor, m: torch.Tensor, v: torch.Tensor, g: torch.Tensor, step: int, lr: float,
                betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.01):
    """One step of the reference's plain-torch Adam on fp32 tensors, fs/optim/adam.py:205-229 (decoupled weight decay
    scaled by lr, eps added to the un-corrected sqrt(v), bias corrections folded into the step size).  ``step`` counts from 1
    (state["step"] after its increment, :206).  In place; returns (p, m, v)."""
    b1, b2 = betas
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    if weight_decay != 0:
        p.add_(p, alpha=-weight_decay * lr)
    p.addcdiv_(m, denom, value=-step_size)
    return p, m, v


def optimizer_update(p, m, v, g_sum, sample_size: float, num_updates: int, lr: float, clip_norm: float,
                     betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.01):
    """What fs/trainer.py:769-795 does with the summed gradient of an update: multiply_grads(1 / sample_size), the norm of
    clip_grad_norm_ (fs/utils.py:341-386: returned for clip_norm 0 too; coefficient max_norm / (norm + 1e-6) clamped to 1),
    no optimizer step when the norm is not finite (:791-793), else Adam with step = num_updates + 1.
    Returns (gnorm, applied)."""
    g = g_sum * (1.0 / sample_size)
    gnorm = float(torch.norm(g, p=2, dtype=torch.float32))
    if not math.isfinite(gnorm):
        return gnorm, False
    if clip_norm > 0:
        g = g * min(1.0, clip_norm / (gnorm + 1e-6))
    adam_update(p, m, v, g, num_updates + 1, lr, betas, eps, weight_decay)
    return gnorm, True


def frame_padding_mask(padding_mask: torch.Tensor, T: int) -> torch.Tensor:
    """Sample-level -> frame-level padding mask, fs/models/wav2vec/wav2vec2.py:560-565 and
    rain/layers/unidirect_w2v2_encoder.py:500-505."""
    extra = padding_mask.size(1) % T
    if extra > 0:
        padding_mask = padding_mask[:, :-extra]
    return padding_mask.view(padding_mask.size(0), T, -1).all(-1)


def streaming_encoder_forward(P: Dict[str, torch.Tensor], source: torch.Tensor, cfg: OracleCfg, *,
                              main_context: int, right_context: int, padding_mask: Optional[torch.Tensor] = None,
                              finished: bool = False, is_infer: bool = False,
                              layer_keep: Optional[List[bool]] = None):
    """Row f1: BlockWiseWav2Vec2Model.forward (rain/layers/unidirect_w2v2_encoder.py:485-531) through
    BlockwiseW2V2TransformerEncoder.forward / extract_features (:245-330): conv extractor, feature
    LayerNorm, post_extract_proj, block-wise encoder with constant contexts, NO masking and NO
    quantizer.  Returns (x [T, B, C], padding_mask [B, T] bool); with ``is_infer and not finished``
    the last ``right_context`` frames are withheld (:326-328).  Dropouts are the identity (eval)."""
    feats = conv_feature_extractor(source, P, cfg)
    if cfg.feature_grad_mult != 1.0:                       # :486-489 GradMultiply: identity fwd, grad * scale
        s = cfg.feature_grad_mult
        feats = feats * s + (feats * (1.0 - s)).detach()
    feats = feats.transpose(1, 2)
    B, T, C0 = feats.shape
    feats = F.layer_norm(feats, (C0,), P["layer_norm.weight"], P["layer_norm.bias"], 1e-5)
    pmf = frame_padding_mask(padding_mask, T) if padding_mask is not None else None
    if "post_extract_proj.weight" in P:
        x = F.linear(feats, P["post_extract_proj.weight"], P["post_extract_proj.bias"])
    else:
        x = feats
    x = blockwise_encoder(x, P, cfg, main_context, right_context, pmf, layer_keep)   # B x T x C
    x = x.transpose(0, 1)
    pad = pmf if pmf is not None else torch.zeros(B, T, dtype=torch.bool)            # :83-84, :319, :324
    if is_infer and not finished and right_context > 0:
        x = x[:-right_context]
        pad = pad[:, :-right_context]
    return x, pad


def online_encoder_forward(P: Dict[str, torch.Tensor], src_tokens: torch.Tensor, src_lengths: torch.Tensor,
                           cfg: OracleCfg, *, main_context: int, right_context: int, finished: bool = False,
                           is_infer: bool = False, prefix: str = "w2v2_model."):
    """OnlineW2V2TransformerEncoder.forward (rain/layers/unidirect_w2v2_encoder.py:587-611): lengths ->
    padding mask (fs/data/data_utils.py:528-532), the twin, optional ``encoder_proj``.  P holds the
    module's own state_dict keys (``w2v2_model.*``, ``encoder_proj.*``)."""
    L = int(src_lengths.max())
    pm = torch.arange(L).view(1, L).expand(src_lengths.numel(), -1) >= src_lengths.view(-1, 1)
    Pw = {k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)}
    x, pad = streaming_encoder_forward(Pw, src_tokens, cfg, main_context=main_context, right_context=right_context,
                                       padding_mask=pm, finished=finished, is_infer=is_infer)
    if "encoder_proj.weight" in P:
        x = F.linear(x, P["encoder_proj.weight"], P["encoder_proj.bias"])
    return x, pad


# ----------------------------------------------------------------------------------
# input side
# ----------------------------------------------------------------------------------
def batch_by_size_vec(indices: np.ndarray, num_tokens_vec: np.ndarray, max_tokens: int, max_sentences: int,
                      bsz_mult: int) -> List[np.ndarray]:
    """fs/data/data_utils_fast.pyx:19-98 restated as a plain Python loop (small cases only).
    Pinned against the reference's own compiled Cython (oracle/_ref, ``make -C oracle ref``) and the
    vectors recorded from it (tests/golden/data_side.npz)."""
    n = len(indices)
    if n == 0:
        return []
    assert max_tokens <= 0 or int(np.max(num_tokens_vec)) <= max_tokens, \
        f"Sentences lengths should not exceed max_tokens={max_tokens}"
    ends = [0] * (n + 1)
    count = 0            # index of the running batch
    start = 0            # where the running batch begins
    tail_max = 0         # longest sample in the not-yet-committed tail
    batch_max = 0        # longest sample in the committed part of the running batch
    for pos in range(n):
        tail_max = max(tail_max, int(num_tokens_vec[pos]))
        new_end = pos + 1
        new_max = max(batch_max, tail_max)
        sentences = new_end - start
        overflow = (max_sentences > 0 and sentences > max_sentences) or (max_tokens > 0 and sentences * new_max > max_tokens)
        mult_ok = sentences < bsz_mult or sentences % bsz_mult == 0
        if overflow:
            if max_tokens > 0 and tail_max * (new_end - ends[count]) > max_tokens:
                count += 1                      # the tail alone overflows: close it before this sample
                ends[count] = pos
                tail_max = int(num_tokens_vec[pos])
            start = ends[count]
            count += 1
            new_max = tail_max
        if overflow or mult_ok:
            ends[count] = new_end
            batch_max = new_max
            tail_max = 0
    if ends[count] != n:
        count += 1
    return np.split(np.asarray(indices), np.asarray(ends[:count], dtype=np.int64))


def ordered_indices(sizes, shuffle: bool = True) -> np.ndarray:
    """RawAudioDataset.ordered_indices, fs/data/audio/raw_audio_dataset.py:214-224: a random permutation
    (numpy global generator) as the tie-break key, sizes as the primary key, longest first."""
    n = len(sizes)
    order = [np.random.permutation(n)] if shuffle else [np.arange(n)]
    order.append(sizes)
    return np.lexsort(order)[::-1]


def collate(sources: List[torch.Tensor], *, pad: bool, max_sample_size: int, normalize: bool = False):
    """RawAudioDataset.collater (raw_audio_dataset.py:123-156) with postprocess's normalisation (:60-72:
    ``F.layer_norm(feats, feats.shape)`` over the whole utterance, before any crop).  Consumes
    ``np.random.randint(0, diff + 1)`` once per utterance longer than the target (:73-81), in order.
    Returns (source [B, target] fp32, padding_mask [B, target] bool or None, crop starts)."""
    if normalize:
        sources = [F.layer_norm(s.float(), s.shape) for s in sources]
    sizes = [len(s) for s in sources]
    target = min(max(sizes), max_sample_size) if pad else min(min(sizes), max_sample_size)
    out = sources[0].new_zeros(len(sources), target)
    pm = torch.zeros(out.shape, dtype=torch.bool) if pad else None
    starts = []
    for i, (src, size) in enumerate(zip(sources, sizes)):
        diff = size - target
        if diff == 0:
            out[i] = src
            starts.append(0)
        elif diff < 0:
            assert pad
            out[i, :size] = src
            pm[i, diff:] = True
            starts.append(0)
        else:
            st = int(np.random.randint(0, diff + 1))
            out[i] = src[st:st + target]
            starts.append(st)
    return out, pm, starts


def init_params(cfg: OracleCfg, seed: int = 1) -> Dict[str, torch.Tensor]:
    """Seeded random parameters with the reference's key names/shapes (SURVEY.md section
    8a 'Parameter inventory') and init distributions (kaiming-normal convs
    wav2vec2.py:724-727; N(0,0.02) linears transformer_sentence_encoder.py:21-53; quantizer
    gumbel_vector_quantizer.py:56-75; mask_emb uniform wav2vec2.py:391-393).  Not the
    reference's RNG stream - only shapes and scales matter for synthetic-weight runs."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    C = cfg.encoder_embed_dim
    in_d = 1
    for i, (dim, k, s) in enumerate(cfg.conv_layers):
        std = math.sqrt(2.0 / (in_d * k))
        P[f"feature_extractor.conv_layers.{i}.0.weight"] = torch.randn(dim, in_d, k, generator=g) * std
        if cfg.conv_bias:
            P[f"feature_extractor.conv_layers.{i}.0.bias"] = torch.zeros(dim)
        if cfg.extractor_mode == "layer_norm" and i < cfg.layer_norm_num:
            P[f"feature_extractor.conv_layers.{i}.2.1.weight"] = torch.ones(dim)
            P[f"feature_extractor.conv_layers.{i}.2.1.bias"] = torch.zeros(dim)
        elif cfg.extractor_mode == "default" and i == 0:
            P[f"feature_extractor.conv_layers.{i}.2.weight"] = torch.ones(dim)
            P[f"feature_extractor.conv_layers.{i}.2.bias"] = torch.zeros(dim)
        in_d = dim
    E = in_d
    fd = cfg.final_dim if cfg.final_dim > 0 else C
    vq = cfg.latent_dim if cfg.latent_dim > 0 else fd

    def lin(name, o, i, std=0.02):
        P[name + ".weight"] = torch.randn(o, i, generator=g) * std
        P[name + ".bias"] = torch.zeros(o)

    def lnp(name, d):
        P[name + ".weight"] = torch.ones(d)
        P[name + ".bias"] = torch.zeros(d)

    P["mask_emb"] = torch.rand(C, generator=g)
    lnp("layer_norm", E)
    lin("post_extract_proj", C, E, 1.0 / math.sqrt(E))
    P["quantizer.vars"] = torch.rand(1, cfg.latent_groups * cfg.latent_vars, vq // cfg.latent_groups, generator=g)
    lin("quantizer.weight_proj", cfg.latent_groups * cfg.latent_vars, E, 1.0)
    lin("project_q", fd, vq, 1.0 / math.sqrt(vq))
    for l in range(cfg.encoder_layers):
        pre = f"encoder.layers.{l}."
        for n in ["q_proj", "k_proj", "v_proj", "out_proj"]:
            lin(pre + "self_attn." + n, C, C)
        lnp(pre + "self_attn_layer_norm", C)
        lin(pre + "fc1", cfg.encoder_ffn_embed_dim, C)
        lin(pre + "fc2", C, cfg.encoder_ffn_embed_dim)
        lnp(pre + "final_layer_norm", C)
    lnp("encoder.layer_norm", C)
    lin("final_proj", fd, C, 1.0 / math.sqrt(C))
    return P
