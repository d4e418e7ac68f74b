"""Python mirror of the reference model API (fs/models/wav2vec/wav2vec_S.py, wav2vec2.py) on top
of the HIP engine.  Same class names, constructor arguments, ``forward`` signature, result
dict, loss hooks and ``state_dict`` keys as the reference, so the fairseq task/criterion and
reference checkpoints drive it unchanged; the modules below hold parameters only - all math
runs in ``engine.py`` through libw2vs.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import engine, host_rng, ops
from ._lib import W2vsError
from .config import (EXTRACTOR_MODE_CHOICES, LAYER_TYPE_CHOICES, MASKING_DISTRIBUTION_CHOICES,  # noqa: F401
                     Wav2VecSConfig)

BF16 = torch.bfloat16


# ------------------------------------------------------------------------------------------------
# parameter containers (names = reference state_dict keys)
# ------------------------------------------------------------------------------------------------
class ConvFeatureExtractionModel(nn.Module):
    """fs/models/wav2vec/wav2vec2.py:702-781.  conv_layers[i] = Sequential(conv, dropout,
    [norm], GELU) so keys read ``conv_layers.i.0.weight`` / ``conv_layers.i.2.1.weight``."""

    def __init__(self, conv_layers: List[Tuple[int, int, int]], dropout: float = 0.0, mode: str = "default",
                 conv_bias: bool = False, layer_norm_num: int = 1):
        super().__init__()
        assert mode in {"default", "layer_norm"}
        self.mode, self.layer_norm_num, self.spec = mode, layer_norm_num, list(conv_layers)
        in_d = 1
        self.conv_layers = nn.ModuleList()
        for i, cl in enumerate(conv_layers):
            assert len(cl) == 3, "invalid conv definition: " + str(cl)
            dim, k, stride = cl
            conv = nn.Conv1d(in_d, dim, k, stride=stride, bias=conv_bias)
            nn.init.kaiming_normal_(conv.weight)
            if mode == "layer_norm" and i < layer_norm_num:
                blk = nn.Sequential(conv, nn.Dropout(p=dropout),
                                    nn.Sequential(nn.Identity(), nn.LayerNorm(dim, elementwise_affine=True), nn.Identity()),
                                    nn.GELU())
            elif mode == "default" and i == 0:
                blk = nn.Sequential(conv, nn.Dropout(p=dropout), nn.GroupNorm(dim, dim, affine=True), nn.GELU())
            else:
                blk = nn.Sequential(conv, nn.Dropout(p=dropout), nn.GELU())
            self.conv_layers.append(blk)
            in_d = dim

    def forward(self, x):
        """fs/models/wav2vec/wav2vec2.py:773-781: B x T waveform -> B x C x T' features, on the HIP kernels (conv0 + LayerNorm /
        GroupNorm + GELU, conv1-6 as channel-last GEMMs).  Forward only: a training step differentiates the extractor inside
        the fused path (engine.backward), this entry serves callers that use the module on its own."""
        if not x.is_cuda:
            raise W2vsError("ConvFeatureExtractionModel runs on an MI355X only (there is no CPU path)")
        with torch.no_grad():
            W = {"feature_extractor.conv_layers." + n: (p if p.dtype == BF16 else p.to(BF16)).contiguous()
                 for n, p in self.conv_layers.named_parameters()}
            y = engine.conv_features(self.spec, self.mode, self.layer_norm_num, W, x.to(BF16).contiguous())   # [B, T', C]
            return ops.transpose2d(y, batch=y.shape[0]).to(x.dtype)                                            # [B, C, T']


class _PosHolder(nn.Module):
    """Stands in for SinusoidalPositionalEmbedding: its only state_dict entry is the
    ``_float_tensor`` buffer (sinusoidal_positional_embedding.py:28)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("_float_tensor", torch.FloatTensor(1).zero_())


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)


class TransformerSentenceEncoderLayer(nn.Module):
    """fs/models/wav2vec/wav2vec2.py:874-978.  Parameters, plus a forward for callers that use a layer on its own."""

    def __init__(self, embedding_dim=768, ffn_embedding_dim=3072, num_attention_heads=8, dropout=0.1,
                 attention_dropout=0.1, activation_dropout=0.1, activation_fn="relu", layer_norm_first=False):
        super().__init__()
        self.self_attn = MultiheadAttention(embedding_dim, num_attention_heads, dropout=attention_dropout)
        self.self_attn_layer_norm = nn.LayerNorm(embedding_dim)
        self.fc1 = nn.Linear(embedding_dim, ffn_embedding_dim)
        self.fc2 = nn.Linear(ffn_embedding_dim, embedding_dim)
        self.final_layer_norm = nn.LayerNorm(embedding_dim)
        self.layer_norm_first = layer_norm_first

    def forward(self, x, self_attn_mask=None, self_attn_padding_mask=None, need_weights=False, att_args=None):
        """wav2vec2.py:921-978, eval semantics (no dropout), forward only, one composite HIP call (w2vs_layer_fwd).
        x: T x B x C.  ``self_attn_mask`` must be None (full attention; block masks are the encoder's business and are
        derived from (T', m, r) there, never materialised); ``self_attn_padding_mask`` B x T bool.  Returns (x, None)."""
        if self_attn_mask is not None or need_weights:
            raise W2vsError("TransformerSentenceEncoderLayer.forward: additive masks / attention weights are not built; "
                            "the block-causal encoder passes (T', m, r) to the attention kernel instead")
        if not x.is_cuda:
            raise W2vsError("TransformerSentenceEncoderLayer runs on an MI355X only (there is no CPU path)")
        with torch.no_grad():
            return engine.single_layer_forward(self, x, self_attn_padding_mask), None


def init_bert_params(module):
    """fs/modules/transformer_sentence_encoder.py:21-53: N(0, 0.02) linears, zero biases."""
    if isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.bias is not None:
            module.bias.data.zero_()


class TransformerEncoder(nn.Module):
    """fs/models/wav2vec/wav2vec2.py:784-871 (parameters only; wav2vec-S replaces pos_conv)."""

    def __init__(self, args):
        super().__init__()
        self.dropout = args.dropout
        self.embedding_dim = args.encoder_embed_dim
        self.layers = nn.ModuleList([
            TransformerSentenceEncoderLayer(self.embedding_dim, args.encoder_ffn_embed_dim, args.encoder_attention_heads,
                                            self.dropout, args.attention_dropout, args.activation_dropout,
                                            args.activation_fn, args.layer_norm_first)
            for _ in range(args.encoder_layers)])
        self.layer_norm_first = args.layer_norm_first
        self.layer_norm = nn.LayerNorm(self.embedding_dim)
        self.layerdrop = args.encoder_layerdrop
        self.apply(init_bert_params)


class BlockwiseTransformerEncoder(TransformerEncoder):
    """fs/models/wav2vec/wav2vec_S.py:335-353."""

    def __init__(self, args):
        super().__init__(args)
        self.pos_type = args.pos_type
        if self.pos_type == "conv":
            raise W2vsError("pos_type='conv' (wav2vec 2.0 conv positions) is not built; wav2vec-S uses 'sin'")
        self.pos_conv = _PosHolder()
        self.required_seq_len_multiple = args.required_seq_len_multiple
        self.context_type = getattr(args, "context_type", "constant")
        self.main_context = getattr(args, "main_context", 16)
        self.right_context = getattr(args, "right_context", 8)


class GumbelVectorQuantizer(nn.Module):
    """fs/modules/gumbel_vector_quantizer.py:11-88 (parameters + temperature schedule)."""

    def __init__(self, dim, num_vars, temp, groups, combine_groups, vq_dim, time_first=True):
        super().__init__()
        self.groups, self.combine_groups, self.input_dim, self.num_vars = groups, combine_groups, dim, num_vars
        assert vq_dim % groups == 0, f"dim {vq_dim} must be divisible by groups {groups} for concatenation"
        if combine_groups:
            raise W2vsError("combine_groups=True is not built (wav2vec 2.0/S use False)")
        var_dim = vq_dim // groups
        self.vars = nn.Parameter(torch.FloatTensor(1, groups * num_vars, var_dim))
        nn.init.uniform_(self.vars)
        self.weight_proj = nn.Linear(dim, groups * num_vars)
        nn.init.normal_(self.weight_proj.weight, mean=0, std=1)
        nn.init.zeros_(self.weight_proj.bias)
        if isinstance(temp, str):
            import ast
            temp = ast.literal_eval(temp)
        assert len(temp) == 3, f"{temp}, {len(temp)}"
        self.max_temp, self.min_temp, self.temp_decay = temp
        self.curr_temp = self.max_temp

    def set_num_updates(self, num_updates):
        self.curr_temp = max(self.max_temp * self.temp_decay ** num_updates, self.min_temp)


def gen_block_attn_mask(x, padding_mask, main_context: int = 1, right_context: int = 0):
    """Same signature and results as the reference helper (wav2vec_S.py:444-489) for callers that
    import it (rain/layers/unidirect_w2v2_encoder.py:18).  The HIP attention never builds this
    mask; this host version exists for API compatibility and tests."""
    if padding_mask is None:
        padding_mask = x.new_zeros((x.size(1), x.size(0)), dtype=torch.bool)
    bsz, seq_len = padding_mask.shape
    lay = host_rng.block_layout(seq_len, main_context, right_context)
    dev = padding_mask.device
    blk = torch.arange(seq_len, device=dev) // main_context
    if right_context == 0:
        masked = blk.unsqueeze(1) < blk.unsqueeze(0)
    else:
        rc_idx = torch.from_numpy(lay.rc_idx).to(dev)
        rc_oob = torch.from_numpy(lay.rc_oob).to(dev)
        rc_blk = torch.arange(seq_len // main_context, device=dev).repeat_interleave(right_context)
        padding_mask = torch.cat((padding_mask, padding_mask.index_select(1, rc_idx) | rc_oob.unsqueeze(0)), dim=1)
        full = torch.cat((blk, rc_blk))
        masked = torch.cat([full.unsqueeze(1) < blk.unsqueeze(0), full.unsqueeze(1).ne(rc_blk.unsqueeze(0))], dim=1)
        x = torch.cat((x, x.index_select(0, rc_idx)), dim=0)
    attn = x.new_zeros(masked.shape).masked_fill(masked, -1e4)
    return x, padding_mask, attn


# ------------------------------------------------------------------------------------------------
# the fused step as one autograd node
# ------------------------------------------------------------------------------------------------
class _HotPath(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, source, padding_mask, mask, features_only, draws, *params):
        cfg = model.cfg
        names = model._param_names
        W, packed = model._weights_for_launch(names, params)
        src = source.detach()
        if src.dtype != BF16:
            src = src.to(BF16)
        src = src.contiguous()
        model._rng_counter += 1
        base = (torch.cuda.initial_seed() * 0x9E3779B97F4A7C15 + model._rng_counter * 0xD1B54A32D192ED03) & ((1 << 64) - 1)
        st = engine.forward(cfg, W, src, training=model.training, mask=mask, features_only=features_only,
                            padding_mask=padding_mask, draws=draws, rng_base=base,
                            tau=float(model.quantizer.curr_temp) if model.quantizer is not None else 1.0,
                            packed=packed, upload_cache=getattr(model, "_upload_cache", None),
                            need_backward=bool(getattr(model, "_grad_mode", True)) and any(ctx.needs_input_grad))
        ctx.st = st
        ctx.model = model
        if model._after_forward is not None and not features_only:
            model._after_forward(st.B * st.M)      # sample_size = number of masked frames: a host integer, known here
        ctx.param_dtypes = [p.dtype for p in params]
        ctx.param_shapes = [tuple(p.shape) for p in params]
        model._last_state = st
        ctx.fused = None
        if features_only:
            return st.out_x
        B, T, C0 = st.B, st.T, st.C0
        fl = model._fused_loss
        if fl is not None and st.qst is not None:
            # criterion.Wav2vecCriterion asked for its own arithmetic on top (one launch instead of ~25 one-element ones):
            # outputs are (loss [1], vec [8]); the logits never become an autograd tensor
            w_ppl, w_pen = fl
            ss = float(st.B * st.M)
            nv = float(model.quantizer.num_vars * model.quantizer.groups)
            loss, vec, dl = ops.infonce_loss(st.logits, st.pen_acc, st.qst.ppl, w_ppl=w_ppl, w_pen=w_pen, num_vars=nv,
                                             pen_norm=1.0 / float(B * T * C0), sample_size=ss,
                                             want_grad=any(ctx.needs_input_grad))
            ctx.fused = (dl, w_pen * ss, -w_ppl * ss / nv)
            ctx.mark_non_differentiable(vec)
            return loss, vec
        pen = st.pen_acc.view(()) / float(B * T * C0)
        prob_ppl, code_ppl = st.qst.ppl[0].clone(), st.qst.ppl[1].clone()
        ctx.mark_non_differentiable(code_ppl)
        if ops.ARENA.active:
            # step arena active: logits and pen are views of the ONE recycled slab; autograd outputs must not be (the
            # slab is written in place all the time), so hand out copies (0.8 MB) and keep the originals for backward
            return st.logits.clone(), pen.clone(), prob_ppl, code_ppl
        return st.logits, pen, prob_ppl, code_ppl

    @staticmethod
    def backward(ctx, *grads):
        st, model = ctx.st, ctx.model
        names = model._param_names
        flat_mode = model._flat is not None
        if flat_mode:
            A = model._flat.arena            # persistent fp32 arena the optimizer / all-reduce work on
        else:
            # the layout (216 names sorted into forward order, offsets) depends on the parameter set only: computed once
            key = tuple(names)
            lay = model._arena_layout if model._arena_layout is not None and model._arena_layout[0] == key else None
            if lay is None:
                A = engine.Arena(engine.grad_shapes(st.cfg, st.W), st.feats.device)
                model._arena_layout = (key, (A.offsets, A.numel), {n: tuple(st.W[n].shape) for n in st.W})
            elif lay[2] != {n: tuple(st.W[n].shape) for n in st.W}:
                A = engine.Arena(engine.grad_shapes(st.cfg, st.W), st.feats.device)
                model._arena_layout = (key, (A.offsets, A.numel), {n: tuple(st.W[n].shape) for n in st.W})
            else:
                A = engine.Arena(None, st.feats.device, layout=lay[1])
        # trainer.TrainStep, first micro-batch of an update: it zeroed the arena EXCEPT engine.wgrad_overwrite_ranges
        ow = bool(flat_mode and getattr(model, "_wgrad_overwrite", False))
        model._wgrad_overwrite = False
        # trainer.TrainStep with update_freq > 1: the transposed weights live as long as the update (FlatParams.wt_cache)
        wtc = getattr(model, "_wt_cache", None) if flat_mode else None
        if st.features_only:
            # the same flags as the pre-training branches: TrainStep zeroed the arena EXCEPT the overwrite ranges whatever the
            # criterion asked the model for (a CTC / fine-tune criterion runs it with features_only=True)
            engine.backward(st, A, d_out=grads[0].to(BF16).contiguous(), on_ready=model._on_grad_ready,
                            overwrite_wgrads=ow, wt_cache=wtc)
        elif ctx.fused is not None:
            dl, c_pen, c_ppl = ctx.fused
            g = grads[0]
            if g is None or dl is None:
                raise W2vsError("fused criterion: no gradient arrived for the loss (or the forward ran without grad)")
            dsc = ops.infonce_loss_bwd(g.detach().float().reshape(1).contiguous(), dl, c_pen, c_ppl)
            engine.backward(st, A, d_logits=dl, d_pen=dsc[0:1], d_prob_ppl=dsc[1:2], on_ready=model._on_grad_ready,
                            overwrite_wgrads=ow, wt_cache=wtc)
        else:
            d_logits, d_pen, d_ppl = grads[0], grads[1], grads[2]
            d_logits = torch.zeros_like(st.logits) if d_logits is None else d_logits.float().contiguous()
            fix = lambda t: None if t is None else t.detach().float().reshape(1).contiguous()  # noqa: E731
            engine.backward(st, A, d_logits=d_logits, d_pen=fix(d_pen), d_prob_ppl=fix(d_ppl),
                            on_ready=model._on_grad_ready, overwrite_wgrads=ow, wt_cache=wtc)
        if flat_mode:
            ctx.st = None
            return (None,) * (6 + len(names))   # gradients stay in the arena (see trainer.FlatParams)
        flat16 = ops.f32_to_bf16(A.flat)
        if ops.ARENA.active:
            # a TrainStep elsewhere in the process left the step arena active: gradients handed to autograd must own
            # their memory (views of the recycled slab trip autograd's view/in-place check and would be overwritten)
            flat16 = flat16.clone()
        dropped = set(range(st.cfg.encoder_layers)) - set(st.kept)
        offs = A.offsets
        fo, no_ext = st.features_only, st.cfg.feature_grad_mult <= 0
        no_mask = st.features_only and st.mask_np is None
        out = []
        for n, dt, shp in zip(names, ctx.param_dtypes, ctx.param_shapes):
            ent = offs.get(n)
            if ent is None:
                out.append(None)
                continue
            li, head, ext, conv_w = _name_kind(n)          # (the same decisions as _is_unused, without re-parsing the name)
            if (li >= 0 and li in dropped) or (fo and head) or (no_ext and ext) or (no_mask and n == "mask_emb"):
                out.append(None)
                continue
            off, numel, ashp = ent
            g = (flat16 if dt == BF16 else A.flat)[off:off + numel].view(ashp)
            if conv_w and len(ashp) == 3:
                g = g.permute(0, 2, 1).contiguous()
            out.append(g if g.dtype == dt else g.to(dt))
        ctx.st = None
        return (None, None, None, None, None, None, *out)


_NAME_KIND = {}


def _name_kind(n):
    """Static facts about a parameter name, parsed once: (encoder layer index or -1, head, extractor, conv weight to permute)."""
    k = _NAME_KIND.get(n)
    if k is None:
        li = int(n.split(".")[2]) if n.startswith("encoder.layers.") else -1
        head = n.startswith("quantizer.") or n.startswith("project_q.") or n.startswith("final_proj.")
        ext = n.startswith("feature_extractor.")
        conv_w = n.startswith("feature_extractor.conv_layers.") and n.endswith(".0.weight")
        k = _NAME_KIND[n] = (li, head, ext, conv_w)
    return k


def _is_unused(n, st, dropped):
    if n.startswith("encoder.layers."):
        li = int(n.split(".")[2])
        return li in dropped           # LayerDrop: the reference leaves these grads None
    if st.features_only and (n.startswith("quantizer.") or n.startswith("project_q.") or n.startswith("final_proj.")):
        return True
    if st.cfg.feature_grad_mult <= 0 and n.startswith("feature_extractor."):
        return True
    if st.features_only and st.mask_np is None and n == "mask_emb":
        return True
    return False


# ------------------------------------------------------------------------------------------------
class Wav2Vec2Model(nn.Module):
    """fs/models/wav2vec/wav2vec2.py:35-699 restricted to what wav2vec-S uses."""

    def __init__(self, cfg: Wav2VecSConfig):
        super().__init__()
        self.cfg = self.args = cfg
        feature_enc_layers = cfg.conv_layers
        self.embed = feature_enc_layers[-1][0]
        self.feature_extractor = ConvFeatureExtractionModel(feature_enc_layers, 0.0, cfg.extractor_mode, cfg.conv_bias,
                                                            cfg.layer_norm_num)
        self.post_extract_proj = (nn.Linear(self.embed, cfg.encoder_embed_dim)
                                  if self.embed != cfg.encoder_embed_dim and not cfg.quantize_input else None)
        self.mask_prob, self.mask_selection, self.mask_other = cfg.mask_prob, cfg.mask_selection, cfg.mask_other
        self.mask_length, self.no_mask_overlap, self.mask_min_space = cfg.mask_length, cfg.no_mask_overlap, cfg.mask_min_space
        self.feature_grad_mult = cfg.feature_grad_mult
        self.n_negatives = cfg.num_negatives
        self.logit_temp = cfg.logit_temp
        final_dim = cfg.final_dim if cfg.final_dim > 0 else cfg.encoder_embed_dim
        self.quantizer = None
        if cfg.quantize_targets:
            vq_dim = cfg.latent_dim if cfg.latent_dim > 0 else final_dim
            self.quantizer = GumbelVectorQuantizer(self.embed, cfg.latent_vars, cfg.latent_temp_tuple, cfg.latent_groups,
                                                   False, vq_dim, True)
            self.project_q = nn.Linear(vq_dim, final_dim)
        else:
            self.project_q = nn.Linear(self.embed, final_dim)
        self.mask_emb = nn.Parameter(torch.FloatTensor(cfg.encoder_embed_dim).uniform_())
        self.encoder = self.build_encoder(cfg)
        self.layer_norm = nn.LayerNorm(self.embed)
        self.final_proj = nn.Linear(cfg.encoder_embed_dim, final_dim)
        self._rng_counter = 0
        self._flat = None                    # trainer.FlatParams when flat storage is active
        self._on_grad_ready = None           # trainer.GradExchange hook: overlap all-reduce with backward
        self._after_forward = None           # trainer.TrainStep hook: called with sample_size once the forward is enqueued
        self._fused_loss = None              # criterion.Wav2vecCriterion: (w_ppl, w_pen) while its forward runs -> loss in the same node
        self._arena_layout = None            # (parameter names, arena offsets) of the per-call gradient arena (non-flat training)
        self._last_state = None
        self._draws = None
        self._np_cache = self._launch_cache = None      # see _named_params_cached / _weights_for_launch
        self.register_load_state_dict_post_hook(lambda m, _inc: m.invalidate_launch_cache())
        self.load_pretrained_model(cfg)

    def build_encoder(self, cfg):
        return BlockwiseTransformerEncoder(cfg)

    # ---- checkpoint interchange -------------------------------------------------------------
    def load_pretrained_model(self, cfg):
        path = cfg.load_pretrained_model_from
        if path:
            state = torch.load(path, map_location="cpu")
            self.load_state_dict(state["model"], strict=False)   # wav2vec2.py:408-415

    def upgrade_state_dict_named(self, state_dict, name):
        return state_dict

    @classmethod
    def build_model(cls, args, task=None):
        cfg = args if isinstance(args, Wav2VecSConfig) else Wav2VecSConfig.from_namespace(args)
        return cls(cfg)

    def set_num_updates(self, num_updates):
        if self.quantizer is not None:
            self.quantizer.set_num_updates(num_updates)

    def max_positions(self):
        return None

    # ---- host draws that callers may inject (parity runs) ---------------------------------------
    def inject_draws(self, draws: Optional[engine.Draws]):
        """Next forward uses these host draws instead of sampling."""
        self._draws = draws

    @property
    def _param_names(self):
        return self._named_params_cached()[0]

    def _named_params_cached(self):
        """(names, parameters) of the module tree, walked once: ``named_parameters()`` costs ~0.4 ms on this model, more than
        the GPU work of a short streaming call.  Dropped when the tree changes (``remove_pretraining_modules``)."""
        c = getattr(self, "_np_cache", None)
        # re-walked whenever gradients are on (0.4 ms is nothing beside a training step, and a fine-tuning script may swap a
        # Parameter object between steps); the no-grad inference calls - the streaming encoder - reuse the list
        if c is None or torch.is_grad_enabled():
            pairs = list(self.named_parameters())
            c = self._np_cache = ([n for n, _ in pairs], [p for _, p in pairs])
        return c

    def _apply(self, fn, *a, **kw):          # .to() / .cuda() / .half(): parameters may be re-homed
        self._np_cache = self._launch_cache = None
        return super()._apply(fn, *a, **kw)

    def _weights_for_launch(self, names, params):
        """name -> bf16 contiguous device tensor for the kernels, plus the launch-side repacks (tap-major conv weights, the
        fused [3E, E] q|k|v weight and bias of every layer).  Outside the flat-parameter trainer these were rebuilt on EVERY
        forward (six transposes and twelve 3.5 MB concatenations: most of a streaming call's launches); they are kept while
        no parameter changed - keyed by (data_ptr, _version, dtype) of every parameter, which any in-place update, load or
        device / dtype move alters."""
        if self._flat is not None:                       # flat storage: the packed forms ARE the storage (trainer.FlatParams)
            packed = self._flat.packed
            W = {}
            for n, p in zip(names, params):
                t = p.detach()
                if t.dtype != BF16:
                    t = t.to(BF16)
                W[n] = t if n in packed else t.contiguous()
            return W, packed
        # Only the no-grad eval path (the streaming encoder) reuses the repacks: an optimizer that writes through ``p.data``
        # (fairseq's Adam does, fs/optim/adam.py:232) changes the values without touching ``p._version``, so during training
        # everything is rebuilt every forward, as before.  ``train()`` / ``eval()``, ``load_state_dict`` and ``.to()`` drop the
        # cache; code that writes ``p.data`` between two inference calls must call ``invalidate_launch_cache()`` itself.
        reuse = not torch.is_grad_enabled() and not self.training
        key = tuple((p.data_ptr(), p._version, p.dtype) for p in params) if reuse else None
        c = getattr(self, "_launch_cache", None)
        if reuse and c is not None and c[0] == key:
            return c[1], c[2]
        W = {}
        for n, p in zip(names, params):
            t = p.detach()
            if t.dtype != BF16:
                t = t.to(BF16)
            W[n] = t.contiguous()
        packed = {}
        if W[names[0]].is_cuda:
            for i in range(1, len(self.cfg.conv_layers)):
                n = "feature_extractor.conv_layers.%d.0.weight" % i
                packed[n] = ops.conv_pack_weight(W[n])
            # fused [3E, E] q|k|v weights and [3E] biases of ALL layers through one multi-tensor copy (a training forward
            # rebuilds them every time: 24 torch.cat launches before)
            pres = ["encoder.layers.%d." % li for li in range(self.cfg.encoder_layers)]
            pres = [pre for pre in pres if pre + "self_attn.q_proj.weight" in W]
            if pres:
                E = W[pres[0] + "self_attn.q_proj.weight"].shape[0]
                dev = W[pres[0] + "self_attn.q_proj.weight"].device
                wall = torch.empty((3 * len(pres), E, E), dtype=BF16, device=dev)
                ball = torch.empty((3 * len(pres), E), dtype=BF16, device=dev)
                src_w = [W[pre + "self_attn.%s_proj.weight" % c] for pre in pres for c in "qkv"]
                src_b = [W[pre + "self_attn.%s_proj.bias" % c] for pre in pres for c in "qkv"]
                torch._foreach_copy_(list(wall.unbind(0)) + list(ball.unbind(0)), src_w + src_b)
                for i, pre in enumerate(pres):
                    packed[pre + "qkv"] = (wall[3 * i:3 * i + 3].view(3 * E, E), ball[3 * i:3 * i + 3].view(3 * E))
        self._launch_cache = (key, W, packed) if reuse else None
        return W, packed

    def invalidate_launch_cache(self):
        self._np_cache = self._launch_cache = None

    def train(self, mode: bool = True):
        self._launch_cache = None
        return super().train(mode)

    # ---- the reference's helper methods, same names ------------------------------------------------
    def sample_negatives(self, y, num):
        """wav2vec2.py:472-526 for the wav2vec-S setting (negatives from the same utterance).  y: B x T x C.  Returns
        (negs [N_neg, B, T, C], neg_idxs [B, N_neg * T]) like the reference; the training step never calls this - its
        InfoNCE kernel gathers by index - it exists for callers of the helper."""
        if self.n_negatives == 0:
            return y.new(0), None
        bsz, tsz, fsz = y.shape
        neg_idxs = host_rng.sample_negative_indices(bsz, num, self.n_negatives)            # same torch.randint draws
        flat = y.reshape(-1, fsz)
        if flat.is_cuda and flat.dtype == BF16:
            negs = ops.gather_rows(flat.contiguous(), neg_idxs.view(-1).to(torch.int32).to(y.device), neg_idxs.numel())
        else:
            negs = flat[neg_idxs.view(-1).to(y.device)]
        negs = negs.view(bsz, num, self.n_negatives, fsz).permute(2, 0, 1, 3)                # to NxBxTxC
        return negs, neg_idxs

    def quantize(self, x):
        """wav2vec2.py:660-665: waveform -> (quantized features B x T x vq_dim, code indices B x T x G), eval semantics."""
        assert self.quantizer is not None
        if not x.is_cuda:
            raise W2vsError("quantize runs on an MI355X only (there is no CPU path)")
        with torch.no_grad():
            W = {n: (p if p.dtype == BF16 else p.to(BF16)).contiguous() for n, p in self.named_parameters()}
            y = engine.conv_features(self.cfg.conv_layers, self.cfg.extractor_mode, self.cfg.layer_norm_num, W,
                                     x.to(BF16).contiguous())                                # [B, T, C0]
            B, T, C0 = y.shape
            feats, _, _, _ = ops.ln_fwd(y, W["layer_norm.weight"], W["layer_norm.bias"])
            G, V = self.cfg.latent_groups, self.cfg.latent_vars
            logits = ops.linear_fwd_f32(feats.view(B * T, C0), W["quantizer.weight_proj.weight"])
            q, qst = ops.quant_fwd(logits, W["quantizer.vars"].view(G * V, -1), G, V, 1.0, False,
                                   bias=W["quantizer.weight_proj.bias"])
            return q.view(B, T, -1), qst.idx.view(B, T, G).long()

    def forward(self, source, padding_mask=None, mask=True, features_only=False):
        if not source.is_cuda:
            raise W2vsError("Wav2VecSModel runs on an MI355X only: move the model and inputs to cuda "
                            "(there is no CPU path; use oracle/ for CPU checks)")
        draws = self._draws if self._draws is not None else engine.Draws()
        self._draws = None
        params = self._named_params_cached()[1]
        self._grad_mode = torch.is_grad_enabled()      # (inside Function.forward grad mode is always off; needs_input_grad ignores it)
        out = _HotPath.apply(self, source, padding_mask, mask, features_only, draws, *params)
        st = self._last_state
        pm = padding_mask
        if padding_mask is not None and st.pad_frames is not None:
            pm = st.pad_frames.to(source.device)
        if features_only:
            return {"x": out, "padding_mask": pm}
        if self._fused_loss is not None and len(out) == 2:
            loss, vec = out
            return {"_fused_loss": loss, "_loss_vec": vec, "sample_size": st.B * st.M, "padding_mask": pm,
                    "prob_perplexity": vec[5], "code_perplexity": vec[6], "features_pen": vec[7],
                    "num_vars": self.quantizer.num_vars * self.quantizer.groups, "temp": self.quantizer.curr_temp}
        logits, pen, prob_ppl, code_ppl = out
        B, M, K = st.B, st.M, st.K
        x = logits.view(B, M, K + 1).permute(2, 0, 1)              # (K+1) x B x M, wav2vec2.py:650
        result = {"x": x, "padding_mask": pm, "features_pen": pen, "_logits_bm": logits}
        if self.quantizer is not None:
            result["prob_perplexity"] = prob_ppl
            result["code_perplexity"] = code_ppl
            result["num_vars"] = self.quantizer.num_vars * self.quantizer.groups
            result["temp"] = self.quantizer.curr_temp
        return result

    def extract_features(self, source, padding_mask, mask=False):
        res = self.forward(source, padding_mask, mask=mask, features_only=True)
        return res["x"], res["padding_mask"]

    def get_logits(self, net_output):
        logits = net_output["x"]
        logits = logits.transpose(0, 2)
        return logits.reshape(-1, logits.size(-1))

    def get_targets(self, sample, net_output, expand_steps=True):
        x = net_output["x"]
        return x.new_zeros(x.size(1) * x.size(2), dtype=torch.long)

    def get_extra_losses(self, net_output):
        pen = []
        if "prob_perplexity" in net_output:
            pen.append((net_output["num_vars"] - net_output["prob_perplexity"]) / net_output["num_vars"])
        if "features_pen" in net_output:
            pen.append(net_output["features_pen"])
        return pen

    def remove_pretraining_modules(self):
        self._np_cache = self._launch_cache = None
        self.quantizer = None
        self.project_q = None
        self.target_glu = None
        self.final_proj = None


class Wav2VecSModel(Wav2Vec2Model):
    """fs/models/wav2vec/wav2vec_S.py:314-332 (registered there as "wav2vec_S")."""
    pass
