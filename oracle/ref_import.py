"""Container-only loader for the REAL reference implementation (test infrastructure).

This file is part of the oracle tooling: it is only ever used by
``tests/golden/gen_golden.py`` and by the in-container ``-m "not gpu"`` tests that
pin ``oracle/w2vs_oracle.py`` against the reference.  Nothing in the product path
(``wav2vec-s_amd/``), ``bench.py``'s GPU leg or the ``-m gpu`` tests imports it, and
it is a no-op when ``/root/reference`` is absent (the GPU box).

Why a loader is needed: ``import fairseq`` raises
``ModuleNotFoundError: omegaconf`` from ``fairseq/dataclass/configs.py:23``.  The
wav2vec-S model files themselves need only torch + numpy, so we register *path-only*
parent packages in ``sys.modules`` (their heavy ``__init__`` never runs) plus three
tiny stubs, and then import the reference's own source files unchanged from where
they lie.  No reference source is copied into this repository.
"""
import dataclasses
import enum
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("W2VS_REFERENCE_ROOT", "/root/reference")
FS = os.path.join(REF_ROOT, "fairseq", "fairseq")


def available() -> bool:
    return os.path.isfile(os.path.join(FS, "models", "wav2vec", "wav2vec_S.py"))


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


_LOADED = None


def load():
    """Return a namespace with the reference's hot-path symbols."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True  # reference tree is read-only

    import torch.nn as nn

    fairseq = _pkg("fairseq", FS)
    modules = _pkg("fairseq.modules", os.path.join(FS, "modules"))
    data = _pkg("fairseq.data", os.path.join(FS, "data"))
    models = _pkg("fairseq.models", os.path.join(FS, "models"))
    w2v = _pkg("fairseq.models.wav2vec", os.path.join(FS, "models", "wav2vec"))
    fairseq.modules, fairseq.data, fairseq.models = modules, data, models
    models.wav2vec = w2v

    # --- stub: fairseq.dataclass (ChoiceEnum, FairseqDataclass) -------------------
    dc = types.ModuleType("fairseq.dataclass")

    def ChoiceEnum(choices):
        return enum.Enum("Choices", {k: k for k in choices})

    @dataclasses.dataclass
    class FairseqDataclass:
        pass

    dc.ChoiceEnum = ChoiceEnum
    dc.FairseqDataclass = FairseqDataclass
    dcu = types.ModuleType("fairseq.dataclass.utils")
    dcu.convert_namespace_to_omegaconf = lambda a: a
    dc.utils = dcu
    sys.modules["fairseq.dataclass"] = dc
    sys.modules["fairseq.dataclass.utils"] = dcu
    fairseq.dataclass = dc

    # --- stub: fairseq.models registry --------------------------------------------
    class BaseFairseqModel(nn.Module):
        def upgrade_state_dict_named(self, state_dict, name):
            return state_dict

        def set_num_updates(self, num_updates):
            for m in self.modules():
                if hasattr(m, "set_num_updates") and m is not self:
                    m.set_num_updates(num_updates)

    def _reg(*a, **k):
        return lambda c: c

    models.BaseFairseqModel = BaseFairseqModel
    models.register_model = _reg
    models.register_model_architecture = _reg

    # --- real reference files ------------------------------------------------------
    imp = importlib.import_module
    imp("fairseq.file_io")
    imp("fairseq.incremental_decoding_utils")
    for name in ["fairseq_dropout", "quant_noise"]:
        imp("fairseq.modules." + name)
    futils = imp("fairseq.utils")  # pulls in modules.multihead_attention itself
    fairseq.utils = futils
    mha = imp("fairseq.modules.multihead_attention")
    modules.MultiheadAttention = mha.MultiheadAttention
    for name, syms in [
        ("fp32_group_norm", ["Fp32GroupNorm"]),
        ("layer_norm", ["Fp32LayerNorm", "LayerNorm"]),
        ("grad_multiply", ["GradMultiply"]),
        ("gumbel_vector_quantizer", ["GumbelVectorQuantizer"]),
        ("same_pad", ["SamePad"]),
        ("transpose_last", ["TransposeLast"]),
        ("sinusoidal_positional_embedding", ["SinusoidalPositionalEmbedding"]),
        ("gelu", ["gelu", "gelu_accurate"]),
        ("fairseq_dropout", ["FairseqDropout"]),
        ("layer_drop", ["LayerDropModuleList"]),
        ("learned_positional_embedding", ["LearnedPositionalEmbedding"]),
        ("positional_embedding", ["PositionalEmbedding"]),
        ("transformer_sentence_encoder_layer", ["TransformerSentenceEncoderLayer"]),
        ("transformer_sentence_encoder", ["TransformerSentenceEncoder"]),
    ]:
        m = imp("fairseq.modules." + name)
        for s in syms:
            setattr(modules, s, getattr(m, s))
    du = imp("fairseq.data.data_utils")
    data.data_utils = du
    w2 = imp("fairseq.models.wav2vec.wav2vec2")
    for s in dir(w2):
        if not s.startswith("_"):
            setattr(w2v, s, getattr(w2, s))
    ws = imp("fairseq.models.wav2vec.wav2vec_S")

    ns = types.SimpleNamespace(
        wav2vec2=w2,
        wav2vec_S=ws,
        data_utils=du,
        utils=futils,
        modules=modules,
        Wav2VecSModel=ws.Wav2VecSModel,
        Wav2VecSConfig=ws.Wav2VecSConfig,
        gen_block_attn_mask=ws.gen_block_attn_mask,
        compute_mask_indices=du.compute_mask_indices,
        GumbelVectorQuantizer=modules.GumbelVectorQuantizer,
        SinusoidalPositionalEmbedding=modules.SinusoidalPositionalEmbedding,
    )
    _LOADED = ns
    return ns


_RAIN = None


def load_rain():
    """The reference's streaming / fine-tune twin:
    ``rain/layers/unidirect_w2v2_encoder.py`` imported unchanged from where it lies.  ``rain`` and
    ``rain.layers`` are registered path-only (their ``__init__`` pulls in the transducer loss and
    SimulEval agents); ``fairseq.options`` / ``fairseq.checkpoint_utils`` need omegaconf and are
    imported-but-unused by that file, so two empty modules stand in for them."""
    global _RAIN
    if _RAIN is not None:
        return _RAIN
    load()
    fairseq = sys.modules["fairseq"]
    for n in ["options", "checkpoint_utils"]:
        if "fairseq." + n not in sys.modules:
            m = types.ModuleType("fairseq." + n)
            sys.modules["fairseq." + n] = m
            setattr(fairseq, n, m)
    d = importlib.import_module("fairseq.data.dictionary")
    sys.modules["fairseq.data"].Dictionary = d.Dictionary
    fe = importlib.import_module("fairseq.models.fairseq_encoder")
    sys.modules["fairseq.models"].FairseqEncoder = fe.FairseqEncoder
    rain_root = os.path.join(REF_ROOT, "rain")
    _pkg("rain", rain_root)
    _pkg("rain.layers", os.path.join(rain_root, "layers"))
    tw = importlib.import_module("rain.layers.unidirect_w2v2_encoder")
    _RAIN = types.SimpleNamespace(
        module=tw,
        BlockWiseWav2Vec2Model=tw.BlockWiseWav2Vec2Model,
        OnlineW2V2TransformerEncoder=tw.OnlineW2V2TransformerEncoder,
        gen_block_atten_mask=tw.gen_block_atten_mask,
    )
    return _RAIN


_DATA = None


def load_data():
    """The reference's input side: ``fs/data/audio/raw_audio_dataset.py`` imported
    unchanged, plus - when ``oracle/_ref`` holds it (``make -C oracle ref``) - the reference's own Cython batcher
    ``data_utils_fast`` compiled from ``fs/data/data_utils_fast.pyx``."""
    global _DATA
    if _DATA is not None:
        return _DATA
    load()
    data = sys.modules["fairseq.data"]
    fd = importlib.import_module("fairseq.data.fairseq_dataset")
    data.FairseqDataset = fd.FairseqDataset
    bw = importlib.import_module("fairseq.data.base_wrapper_dataset")
    data.BaseWrapperDataset = bw.BaseWrapperDataset
    _pkg("fairseq.data.audio", os.path.join(FS, "data", "audio"))
    rad = importlib.import_module("fairseq.data.audio.raw_audio_dataset")
    fast = None
    ref_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
    if os.path.isdir(ref_dir):
        sys.path.insert(0, ref_dir)
        try:
            fast = importlib.import_module("data_utils_fast")
        except ImportError:
            fast = None
        finally:
            sys.path.remove(ref_dir)
    if fast is not None:
        sys.modules["fairseq.data.data_utils_fast"] = fast     # what fs/data/data_utils.py:311-315 imports
    _DATA = types.SimpleNamespace(raw_audio_dataset=rad, RawAudioDataset=rad.RawAudioDataset,
                                  FileAudioDataset=rad.FileAudioDataset, data_utils=sys.modules["fairseq.data.data_utils"],
                                  data_utils_fast=fast)
    return _DATA


_JOINER = None


def load_joiner():
    """The reference's CAAT joint network: ``ExpandMultiheadAttention``,
    ``TransformerJointerLayer`` and ``MHAJointNet`` of ``rain/layers/attention_transducer.py:590-852``.

    That file cannot be imported as a module here: its header pulls in ``omegaconf``, the CUDA-only ``warprnnt_pytorch`` and
    the whole fairseq Transformer model family (``ModuleNotFoundError`` - ordinary Python errors, nothing was refused).  The
    three classes themselves need only torch and four fairseq helpers that DO import (``FairseqDropout``, ``LayerNorm``,
    ``fairseq.utils``, ``with_incremental_state``), so their source lines are read from the reference file where it lies and
    executed unchanged in a namespace holding exactly those real dependencies.  Nothing is copied into the repository."""
    global _JOINER
    if _JOINER is not None:
        return _JOINER
    load()
    import argparse
    import math
    import random
    from typing import Any, Dict, List, Optional, Tuple

    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    idu = importlib.import_module("fairseq.incremental_decoding_utils")
    fm = sys.modules["fairseq.modules"]
    path = os.path.join(REF_ROOT, "rain", "layers", "attention_transducer.py")
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("class ExpandMultiheadAttention")) - 1   # its decorator line
    stop = next(i for i, l in enumerate(lines) if l.startswith("class TransducerMHADecoder"))
    assert lines[start].startswith("@with_incremental_state")
    ns = dict(torch=torch, nn=nn, F=F, math=math, random=random, Tensor=torch.Tensor, Dict=Dict, List=List, Optional=Optional,
              Tuple=Tuple, Any=Any, Namespace=argparse.Namespace, utils=sys.modules["fairseq"].utils,
              FairseqDropout=fm.FairseqDropout, LayerNorm=fm.LayerNorm, with_incremental_state=idu.with_incremental_state)
    code = compile("\n" * start + "\n".join(lines[start:stop]), path, "exec")       # keeps the reference's line numbers
    exec(code, ns)
    _JOINER = types.SimpleNamespace(MHAJointNet=ns["MHAJointNet"], TransformerJointerLayer=ns["TransformerJointerLayer"],
                                    ExpandMultiheadAttention=ns["ExpandMultiheadAttention"])
    return _JOINER


_HEAD = None


def load_transducer_out():
    """The reference's CAAT loss head ``TransducerOut`` (``rain/layers/attention_transducer.py:289-456``), executable on the CPU
    at ``delay_scale = 0``.

    The class source is read from the reference file where it lies and executed unchanged (as ``load_joiner`` does for the
    joint network), together with ``label_smoothed_nll_loss`` (``fs/criterions/label_smoothed_cross_entropy.py:33-50``, whose
    module needs omegaconf - an ordinary ModuleNotFoundError).  Its one dependency that exists only for CUDA,
    ``warprnnt_pytorch.DelayTLoss``, is bound to the reference's OWN CPU transducer compiled into
    ``oracle/_ref/libwarprnnt_cpu.so`` (``make -C oracle ref``; wt/src/rnnt_entrypoint.cpp + cpu_rnnt.h): same constructor and
    ``forward(acts, labels, act_lens, label_lens) -> (total, rnnt, delay)`` (delay_transducer.py:45-178), log-softmax applied
    in front as warprnnt_pytorch/rnnt.py:67-68 does for the CPU path.  The CPU transducer has no delay terms, so the binding
    accepts ``delay_scale == 0`` only and reports the delay cost as 0: everything but the delay term itself is the
    reference's arithmetic."""
    global _HEAD
    if _HEAD is not None:
        return _HEAD
    load()
    import numpy as np
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from torch import autograd
    import rnnt_oracle as R
    assert R.RefCpuRnnt.available(), "build oracle/_ref first: make -C oracle ref"
    cpu = R.RefCpuRnnt()

    class _RefRnnt(autograd.Function):
        @staticmethod
        def forward(ctx, log_probs, labels, act_lens, label_lens, blank):
            costs, g = cpu.loss_and_logprob_grads(log_probs.detach().numpy(), labels.numpy(), act_lens.numpy(),
                                                  label_lens.numpy(), blank)
            ctx.g = torch.from_numpy(np.ascontiguousarray(g))
            return torch.from_numpy(costs.astype(np.float32)).sum()

        @staticmethod
        def backward(ctx, grad_out):
            return ctx.g * grad_out, None, None, None, None

    class DelayTLoss(nn.Module):              # constructor of delay_transducer.py:149-166
        def __init__(self, blank=0, delay_scale=1.0, temperature=1.0, reduction="sum", delay_func="zero"):
            super().__init__()
            assert delay_scale == 0 and temperature == 1.0 and reduction == "sum", "the CPU transducer has no delay terms"
            self.blank = blank

        def forward(self, acts, labels, act_lens, label_lens):
            total = _RefRnnt.apply(F.log_softmax(acts.float(), dim=-1), labels, act_lens, label_lens, self.blank)
            return total, total.detach(), torch.zeros(())

    def _lines(path, first_prefix, stop_prefixes):
        lines = open(path).read().split("\n")
        start = next(i for i, l in enumerate(lines) if l.startswith(first_prefix))
        stop = next(i for i in range(start + 1, len(lines)) if any(lines[i].startswith(sp) for sp in stop_prefixes))
        return compile("\n" * start + "\n".join(lines[start:stop]), path, "exec")     # keeps the reference's line numbers

    ns = dict(torch=torch, nn=nn, F=F, Tensor=torch.Tensor, autograd=autograd, DelayTLoss=DelayTLoss)
    exec(_lines(os.path.join(REF_ROOT, "fairseq", "fairseq", "criterions", "label_smoothed_cross_entropy.py"),
                "def label_smoothed_nll_loss", ("@register_criterion", "class ")), ns)
    exec(_lines(os.path.join(REF_ROOT, "rain", "layers", "attention_transducer.py"), "class TransducerOut", ("class ", "def ")), ns)
    _HEAD = types.SimpleNamespace(TransducerOut=ns["TransducerOut"], DelayTLoss=DelayTLoss)
    return _HEAD


_OPTIM = None


def load_optim():
    """The reference's optimizer pieces, executable on the CPU:

    * ``Adam`` - ``fs/optim/adam.py:103-229``, the plain-torch class FairseqAdam falls back to (and that FP16Optimizer drives
      on the fp32 master copy, fs/optim/fp16_optimizer.py:205-218).  Its MODULE needs omegaconf (an ordinary
      ModuleNotFoundError), the class itself only torch + math: its source lines are read from the file where it lies and
      executed unchanged, as ``load_joiner`` does.
    * ``clip_grad_norm_`` - ``fs/utils.py:341-386``: ``fairseq.utils`` imports as a real module (``load()``).
    * ``PolynomialDecayLRSchedule`` - ``fs/optim/lr_scheduler/polynomial_decay_schedule.py:40-89``, class source executed
      unchanged over a minimal ``FairseqLRScheduler`` base (cfg / optimizer / best: what fairseq_lr_scheduler.py:12-19 sets;
      the real base's module needs omegaconf through ``fairseq.optim``) and an optimizer handle with ``set_lr`` / ``get_lr``.
    Nothing is copied into the repository."""
    global _OPTIM
    if _OPTIM is not None:
        return _OPTIM
    ns0 = load()
    import math

    import torch

    def _lines(path, first_prefix, stop_prefixes):
        lines = open(path).read().split("\n")
        start = next(i for i, l in enumerate(lines) if l.startswith(first_prefix))
        stop = next((i for i in range(start + 1, len(lines)) if any(lines[i].startswith(sp) for sp in stop_prefixes)),
                    len(lines))
        return compile("\n" * start + "\n".join(lines[start:stop]), path, "exec")     # keeps the reference's line numbers

    ns = dict(torch=torch, math=math)
    exec(_lines(os.path.join(FS, "optim", "adam.py"), "class Adam(torch.optim.Optimizer)", ("class ", "def ", "@")), ns)

    class FairseqLRScheduler(object):             # fs/optim/lr_scheduler/fairseq_lr_scheduler.py:12-19
        def __init__(self, cfg, optimizer):
            super().__init__()
            self.cfg = cfg
            self.optimizer = optimizer
            self.best = None

    ns2 = dict(FairseqLRScheduler=FairseqLRScheduler, PolynomialDecayLRScheduleConfig=object)   # the annotation only
    exec(_lines(os.path.join(FS, "optim", "lr_scheduler", "polynomial_decay_schedule.py"),
                "class PolynomialDecayLRSchedule(", ("class ", "def ", "@")), ns2)

    class LrHandle:                               # FairseqOptimizer.get_lr / set_lr (fs/optim/fairseq_optimizer.py:78-85)
        def __init__(self, opt):
            self.opt = opt

        def get_lr(self):
            return self.opt.param_groups[0]["lr"]

        def set_lr(self, lr):
            for g in self.opt.param_groups:
                g["lr"] = lr

    _OPTIM = types.SimpleNamespace(Adam=ns["Adam"], clip_grad_norm_=ns0.utils.clip_grad_norm_,
                                   PolynomialDecayLRSchedule=ns2["PolynomialDecayLRSchedule"], LrHandle=LrHandle)
    return _OPTIM


def make_cfg(ref, **overrides):
    """Wav2VecSConfig with the base yaml's model overrides
    (fairseq/examples/wav2vec/config/pretraining/wav2vec-S_base_librispeech.yaml:50-77)."""
    cfg = ref.Wav2VecSConfig()
    base = dict(
        quantize_targets=True,
        extractor_mode="layer_norm",
        final_dim=256,
        encoder_layerdrop=0.05,
        dropout_input=0.1,
        dropout_features=0.1,
        encoder_embed_dim=768,
        feature_grad_mult=0.1,
        main_context=16,
        right_context=8,
        context_type="sampling",
        pos_type="sin",
        mask_length=10,
        mask_prob=0.65,
        mask_selection="static",
        conv_feature_layers="[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] * 2",
    )
    base.update(overrides)
    for k, v in base.items():
        assert hasattr(cfg, k), k
        setattr(cfg, k, v)
    return cfg
