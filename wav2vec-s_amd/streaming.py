"""Streaming / fine-tune twin of the hot path.

Host-side mirror of the reference's ``rain/layers/unidirect_w2v2_encoder.py``:

* ``gen_block_atten_mask``           (:67-115)  - API helper only; the HIP attention derives the block structure
                                                 from (T', m, r) and never builds the N x N mask
* ``BlockWiseWav2Vec2Model``         (:443-531) - conv extractor -> feature LayerNorm -> post_extract_proj ->
                                                 dropout_input -> block-wise encoder; no masking, no quantizer
* ``OnlineW2V2TransformerEncoder``   (:534-676) - checkpoint loading, ``freeze_finetune_updates``, optional
                                                 ``encoder_proj``, fairseq encoder-out dictionary, ``reorder_encoder_out``

Everything that computes runs through the same HIP kernels as pre-training (``engine.forward`` with
``features_only=True``): same names, argument meaning and return structure as the reference classes, T x B x C
output, frame-level padding mask, right-context trimming while a stream is unfinished.  There is no CPU path.
"""
import argparse
import contextlib
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from ._lib import W2vsError
from .config import Wav2VecSConfig
from .model import Wav2Vec2Model, gen_block_attn_mask

BF16 = torch.bfloat16

# rain/layers/unidirect_w2v2_encoder.py:679-745 - defaults the twin fills in for absent arguments (they differ from
# the pre-training dataclass: main_context 8 / right_context 4, no quantizer).
_TWIN_DEFAULTS = dict(
    extractor_mode="default", encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072,
    encoder_attention_heads=12, activation_fn="gelu", dropout=0.1, attention_dropout=0.1, activation_dropout=0.0,
    final_dim=0, layer_norm_first=False, encoder_layerdrop=0.0,
    conv_feature_layers="[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] + [(512,2,2)]",
    logit_temp=0.1, quantize_targets=False, quantize_input=False, same_quantizer=False, feature_grad_mult=1.0,
    latent_vars=320, latent_groups=2, latent_dim=0, mask_length=10, mask_prob=0.65, mask_selection="static",
    mask_other=0, no_mask_overlap=False, mask_min_space=1, mask_channel_length=10, mask_channel_prob=0,
    mask_channel_selection="static", mask_channel_other=0, no_mask_channel_overlap=False, mask_channel_min_space=1,
    dropout_input=0, dropout_features=0, num_negatives=100, negatives_from_everywhere=False,
    cross_sample_negatives=0, codebook_negatives=0, conv_pos=128, conv_pos_groups=16, target_glu=False,
    conv_bias=False, main_context=8, right_context=4, required_seq_len_multiple=2, simul_mode=None,
)


def base_architecture(args):
    """Fill absent attributes in place, as the reference's function of the same name does."""
    for k, v in _TWIN_DEFAULTS.items():
        if not hasattr(args, k):
            setattr(args, k, v)
    args.latent_temp = str(getattr(args, "latent_temp", "(2,0.5,0.999995)"))
    return args


def gen_block_atten_mask(x: Tensor, padding_mask: Optional[Tensor], main_context: int = 1, right_context: int = 0,
                         attn_mask_value: float = -1e4):
    """Reference spelling and signature (:67-115).  Same block structure as the pre-training helper."""
    x, padding_mask, attn = gen_block_attn_mask(x, padding_mask, main_context, right_context)
    if attn_mask_value != -1e4:
        attn = attn * (attn_mask_value / -1e4)
    return x, padding_mask, attn


def lengths_to_padding_mask(lens: Tensor) -> Tensor:
    """fs/data/data_utils.py lengths_to_padding_mask: True where position >= length."""
    bsz, max_lens = lens.size(0), int(torch.max(lens).item())
    return torch.arange(max_lens, device=lens.device).view(1, max_lens).expand(bsz, -1) >= lens.view(bsz, 1)


def _encoder_out(x, pad):
    return {
        "encoder_out": [x],                  # T x B x C
        "encoder_padding_mask": [pad],       # B x T
        "encoder_embedding": [],
        "encoder_states": [],
        "src_tokens": [],
        "src_lengths": [],
        "dec1_state": [],
        "dec1_padding_mask": [],
    }


class BlockWiseWav2Vec2Model(Wav2Vec2Model):
    """rain/layers/unidirect_w2v2_encoder.py:443-531.  Parameters and state_dict keys are those of the pre-training
    model (the quantizer / projection heads are built when the checkpoint's arguments ask for them, as in the
    reference, so a pre-training checkpoint loads without key filtering)."""

    def __init__(self, cfg):
        if not isinstance(cfg, Wav2VecSConfig):
            cfg = Wav2VecSConfig.from_namespace(base_architecture(cfg))
        cfg.context_type = "constant"        # the twin has no context sampling (:305-307)
        super().__init__(cfg)
        self._upload_cache = {}              # device copies of the per-shape index arrays (engine.forward, upload_cache)

    @classmethod
    def build_model(cls, args, task=None):
        return cls(args)

    # A streaming call is ~150 small kernels whose GPU time is their dispatch latency (0.9 / 1.4 / 1.8 ms at 2 / 10 / 30 s
    # prefixes, nearly flat in the prefix length).  For a fixed input shape the whole launch sequence - index upload, extractor,
    # prologue, twelve composite layer calls, gather - is the same every call, so inference calls without a padding mask can be
    # captured per shape into a HIP graph and replayed (W2VS_STREAM_GRAPH=0, or ``graph_calls = False``, runs them eagerly).
    # A real streaming session re-encodes a GROWING prefix: every call of an utterance has a new shape, and a capture costs two
    # warm-up forwards, a capture, an instantiate and the activations it holds.  So a shape is captured only once it has been
    # seen ``graph_after`` times eagerly (fixed chunk schedules make later utterances repeat the shapes of the first; a shape
    # met once never pays for a capture), graphs are evicted least-recently-used, and both their number and the bytes they
    # hold are bounded.
    # Measured (profiles/round5_stream_encoder_bench.json): a replay is no faster than the eager call on the GPU (1.39 ms either
    # way at a 10 s prefix - the chain of ~150 dependent kernels is what a call costs), a capture costs ~4 ms and a graph holds
    # ~170 MB of activations.  What a replay saves is the HOST's issue time (~1.1 ms of one core per call).  So replay is
    # opt-in (``graph_calls = True`` or W2VS_STREAM_GRAPH=1), for deployments whose host thread is the scarce resource.
    graph_calls = os.environ.get("W2VS_STREAM_GRAPH", "0") == "1"
    graph_after = 2                 # eager calls of a shape before it is captured
    max_graphs = 128                # captured shapes kept (30 s at 320 ms chunks = 94 shapes)
    max_graph_bytes = 32 << 30      # device bytes the kept graphs may hold (of 288 GB; ~170 MB each at a 10 s prefix)

    def _graph_key(self):
        return tuple((p.data_ptr(), p._version, p.dtype) for p in self._named_params_cached()[1])

    def graph_stats(self):
        """{"hits", "misses", "captures", "graphs", "bytes"} of the per-shape graph cache (tools/bench_stream.py reports them)."""
        st = self.__dict__.setdefault("_gstats", {"hits": 0, "misses": 0, "captures": 0})
        graphs = self.__dict__.get("_graphs") or {}
        ents = [v for k, v in graphs.items() if isinstance(k, tuple)]
        return dict(st, graphs=len(ents), bytes=sum(e[4] for e in ents))

    def _forward_graphed(self, source):
        """-> (x [B, T, C], state) through the captured graph of this shape, or None while the shape is not worth one yet."""
        import collections
        graphs = self.__dict__.setdefault("_graphs", collections.OrderedDict())
        seen = self.__dict__.setdefault("_gseen", collections.OrderedDict())
        stats = self.__dict__.setdefault("_gstats", {"hits": 0, "misses": 0, "captures": 0})
        wkey = self._graph_key()
        # a graph bakes in the weights' addresses AND the launch-side repacks of the call it was captured from: recapture
        # everything when a parameter changed (in place, reloaded, moved: its (data_ptr, _version, dtype)) or when the repack
        # cache was dropped (load_state_dict, train() / eval(), .to(), invalidate_launch_cache() after a write through .data)
        if graphs and (graphs.get("_wkey") != wkey or graphs.get("_lc") is not self._launch_cache or self._launch_cache is None):
            graphs.clear()
        key = (tuple(source.shape), source.dtype, source.device.index)
        ent = graphs.get(key)
        if ent is None:
            stats["misses"] += 1
            n = seen.pop(key, 0) + 1
            seen[key] = n
            while len(seen) > 4096:
                seen.popitem(last=False)
            if n <= self.graph_after:
                return None                              # run this call eagerly
            seen.pop(key, None)
            cur = torch.cuda.current_stream()
            before = torch.cuda.memory_allocated()
            static_in = source.clone()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):                # warm-up: lazy allocations inside the library, the launch-weight
                super().forward(static_in, None, mask=False, features_only=True)   # cache and the pinned index buffer exist
            cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                res = super().forward(static_in, None, mask=False, features_only=True)
            # the state keeps the pinned index buffer the graph's copy node reads from, and every intermediate, alive
            held = max(0, torch.cuda.memory_allocated() - before)
            ent = (g, static_in, res["x"], self._last_state, held)
            graphs[key] = ent
            graphs["_wkey"], graphs["_lc"] = wkey, self._launch_cache
            stats["captures"] += 1
            shapes = [k for k in graphs if isinstance(k, tuple)]
            total = sum(graphs[k][4] for k in shapes)
            while len(shapes) > 1 and (len(shapes) > self.max_graphs or total > self.max_graph_bytes):
                old = shapes.pop(0)                      # least recently used first
                total -= graphs.pop(old)[4]
        else:
            stats["hits"] += 1
            graphs.move_to_end(key)
        g, static_in, x, st, _ = ent
        static_in.copy_(source)
        g.replay()
        self._last_state = st
        return x.clone(), st                             # the graph's output buffer is rewritten by the next replay

    def forward(self, source, padding_mask=None, incremental_state=None, finished=False, is_infer=False):
        """source [B, L] waveform, padding_mask [B, L] bool (True = padding).  Returns the fairseq encoder-out
        dictionary: ``encoder_out`` [T, B, C] and ``encoder_padding_mask`` [B, T].  With ``is_infer`` and not
        ``finished`` the last ``right_context`` frames are withheld (:326-328): they have not seen their own right
        context yet and are emitted by a later call on the longer prefix."""
        if (self.graph_calls and padding_mask is None and source.is_cuda and source.dtype == BF16 and not self.training
                and not torch.is_grad_enabled() and self._draws is None and not ops.ARENA.active):
            got = self._forward_graphed(source)
        else:
            got = None
        if got is not None:
            (x, st), pad = got, None
        else:
            res = super().forward(source, padding_mask, mask=False, features_only=True)
            st = self._last_state
            x, pad = res["x"], res["padding_mask"]
        x = x.transpose(0, 1)                                            # B x T x C -> T x B x C (:311, :321)
        if pad is None:
            pad = torch.zeros(st.B, st.T, dtype=torch.bool, device=source.device)   # :83-84
        r = self.cfg.right_context
        if is_infer and not finished and r > 0:
            x = x[:-r]
            pad = pad[:, :-r]
        return _encoder_out(x, pad)


class _HipLinear(torch.autograd.Function):
    """nn.Linear on the HIP GEMMs (forward, data gradient, weight/bias gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        shp = x.shape
        x2 = x.detach().to(BF16).reshape(-1, shp[-1]).contiguous()
        w = weight.detach().to(BF16).contiguous()
        b = bias.detach().to(BF16).contiguous() if bias is not None else None
        y = ops.linear_fwd(x2, w, b)
        if ops.ARENA.active:
            y = y.clone()
        ctx.save_for_backward(x2, w)
        ctx.meta = (shp, x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        shp, xdt, wdt, bdt = ctx.meta
        dy2 = dy.to(BF16).reshape(-1, w.shape[0]).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_dgrad(dy2, ops.transpose2d(w))
            dx = (dx.clone() if ops.ARENA.active else dx).view(shp).to(xdt)
        if ctx.needs_input_grad[1] or (bdt is not None and ctx.needs_input_grad[2]):
            dw32 = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
            db32 = torch.zeros(w.shape[0], dtype=torch.float32, device=w.device) if bdt is not None else None
            ops.linear_wgrad(dy2, x2, dw32, 1.0, db32)
            dw = dw32.to(wdt)
            db = db32.to(bdt) if bdt is not None else None
        return dx, dw, db


class HipLinear(nn.Linear):
    """Same parameters / state_dict keys as ``nn.Linear``; computes on the GPU library only."""

    def forward(self, x):
        if not x.is_cuda:
            raise W2vsError("HipLinear runs on an MI355X only (there is no CPU path)")
        return _HipLinear.apply(x, self.weight, self.bias)


class OnlineW2V2TransformerEncoder(nn.Module):
    """rain/layers/unidirect_w2v2_encoder.py:534-676: the encoder the streaming ST/ASR models build
    (rain/models/unidirect_w2v2_transformer.py:64, rain/models/w2v2_transducer.py:210).

    ``args``: ``w2v2_model_path`` (a fairseq checkpoint: ``{"args" | "cfg": {"model": ...}, "model": state_dict}``),
    ``main_context``, ``right_context``, ``use_linear_layer``, ``encoder_embed_dim``, ``freeze_finetune_updates``."""

    def __init__(self, args):
        super().__init__()
        self.main_context = args.main_context
        self.right_context = args.right_context
        ckpt = torch.load(args.w2v2_model_path, map_location="cpu", weights_only=False)
        if ckpt.get("args") is None:
            w2v2_args = argparse.Namespace(**ckpt["cfg"]["model"])               # :544-545
        else:
            w2v2_args = ckpt["args"]                                            # :546-549 (old checkpoints)
            w2v2_args.extractor_mode = "layer_norm"
            w2v2_args.pos_type = "sin"
        w2v2_args.main_context = args.main_context
        w2v2_args.right_context = args.right_context
        if not hasattr(w2v2_args, "load_pretrained_model_from"):
            w2v2_args.load_pretrained_model_from = None
        self.w2v2_model = BlockWiseWav2Vec2Model.build_model(w2v2_args, task=None)
        self.w2v2_model.load_state_dict(ckpt["model"], strict=False)
        self.use_linear_layer = args.use_linear_layer
        self.encoder_proj = None
        if self.use_linear_layer and w2v2_args.encoder_embed_dim != args.encoder_embed_dim:
            self.encoder_proj = HipLinear(w2v2_args.encoder_embed_dim, args.encoder_embed_dim)
        self.freeze_finetune_updates = getattr(args, "freeze_finetune_updates", -1)
        self.num_updates = 0

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates
        self.w2v2_model.set_num_updates(num_updates)

    @property
    def init_frames(self):
        return self.main_context + self.right_context

    @property
    def step_frames(self):
        return self.main_context

    def forward(self, src_tokens, src_lengths, incremental_state=None, finished=False, is_infer=False):
        padding_mask = lengths_to_padding_mask(src_lengths)
        ft = self.freeze_finetune_updates <= self.num_updates                   # :590
        with torch.no_grad() if not ft else contextlib.ExitStack():
            output = self.w2v2_model(src_tokens, padding_mask, incremental_state, finished, is_infer)
        if self.use_linear_layer and self.encoder_proj is not None:
            x = self.encoder_proj(output["encoder_out"][0])
            return _encoder_out(x, output["encoder_padding_mask"][0])
        return output

    def forward_torchscript(self, net_input: Dict[str, Tensor]):
        return self.forward(src_tokens=net_input["src_tokens"], src_lengths=net_input["src_lengths"])

    def max_positions(self):
        return None

    @torch.jit.unused
    def reorder_encoder_out(self, encoder_out: Dict[str, List[Tensor]], new_order):
        """:620-676 - beam reordering of every populated entry (time-major entries along dim 1, batch-major along 0)."""
        time_major = {"encoder_out", "dec1_state"}
        out = {}
        for key, val in encoder_out.items():
            if key == "encoder_states":
                out[key] = [s.index_select(1, new_order) for s in val]
            elif len(val) == 0:
                out[key] = []
            else:
                out[key] = [val[0].index_select(1 if key in time_major else 0, new_order)]
        return out
