"""Mirror of fairseq's ``wav2vec`` criterion (fs/criterions/wav2vec_criterion.py:36-157) for the
InfoNCE configuration wav2vec-S trains with: same constructor arguments, same
``forward(model, sample) -> (loss, sample_size, logging_output)`` contract and logging keys.
The cross entropy and the accuracy counters run in one HIP kernel (w2vs_ce_rows)."""
import math

import torch

from . import ops


class _Meters:
    """Minimal stand-in for ``fairseq.metrics`` (the logging registry is outside SURVEY section 8): ``log_scalar`` keeps
    weighted sums, ``log_derived`` a function of them; ``reduce_metrics`` writes into one of these when the caller passes
    none of its own.  ``get(k)`` = the weighted average (or plain sum for weight 0), as fairseq's AverageMeter reports it."""

    def __init__(self):
        self.sums, self.weights, self.rounds, self.derived = {}, {}, {}, {}

    def log_scalar(self, key, value, weight=1, round=None, priority=10):
        self.sums[key] = self.sums.get(key, 0.0) + float(value) * (weight if weight else 1)
        self.weights[key] = self.weights.get(key, 0.0) + (weight if weight else 0)
        self.rounds[key] = round

    def log_derived(self, key, fn, priority=20):
        self.derived[key] = fn

    def get(self, key):
        if key in self.derived:
            class _M:          # what the derived lambdas read: meters[k].sum
                def __init__(s_, v): s_.sum = v
            return self.derived[key]({k: _M(v) for k, v in self.sums.items()})
        w = self.weights.get(key, 0.0)
        v = self.sums[key] / w if w else self.sums[key]
        r = self.rounds.get(key)
        return round(v, r) if r is not None else v


def _safe_round(x, nd):
    return round(float(x), nd)


class _CrossEntropyTarget0(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        out3, dl = ops.ce_rows(logits.contiguous(), want_grad=True)
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(out3)
        return out3[0].clone(), out3

    @staticmethod
    def backward(ctx, g, _):
        (dl,) = ctx.saved_tensors
        return dl * g


class Wav2vecCriterion:
    def __init__(self, task=None, infonce=False, loss_weights=None, log_keys=None):
        if not infonce:
            raise NotImplementedError("only the InfoNCE form (infonce=True) is on the wav2vec-S path")
        self.infonce = infonce
        self.loss_weights = loss_weights
        self.log_keys = [] if log_keys is None else log_keys
        self.training = True
        self.fuse = True          # False: always compose the loss from framework ops (the reference's own sequence)

    def __call__(self, model, sample, reduce=True, sync_logging=True):
        return self.forward(model, sample, reduce, sync_logging)

    def forward(self, model, sample, reduce=True, sync_logging=True):
        """sync_logging=False keeps the logged scalars as device tensors (no .item() host syncs);
        the reference always syncs (wav2vec_criterion.py:110, 128, 133, 151)."""
        # Fast path: wav2vec-S's own configuration (two extra losses, both weighted, quantizer on) on this package's model -
        # the cross entropy, the weighting below and their backward run inside the model's autograd node (w2vs_infonce_loss).
        fused = (self.loss_weights is not None and len(self.loss_weights) == 2 and all(c != 0 for c in self.loss_weights)
                 and getattr(model, "quantizer", None) is not None and hasattr(model, "_fused_loss") and self.fuse)
        if fused:
            model._fused_loss = (float(self.loss_weights[0]), float(self.loss_weights[1]))
        try:
            net_output = model(**sample["net_input"])
        finally:
            if fused:
                model._fused_loss = None
        if "_fused_loss" in net_output:
            return self._fused_outputs(net_output, sample, sync_logging)
        if "_logits_bm" in net_output:
            logits = net_output["_logits_bm"]           # same rows as get_logits, (b, m) order; the sum is order free
        else:
            logits = model.get_logits(net_output).float()
        loss, out3 = _CrossEntropyTarget0.apply(logits)
        sample_size = logits.shape[0]
        losses = [loss.detach().clone()]
        if self.loss_weights is not None:
            extra_losses = model.get_extra_losses(net_output)
            if torch.is_tensor(extra_losses):
                extra_losses = [extra_losses]
            if len(self.loss_weights) == 1 and len(extra_losses) != 1:
                self.loss_weights = [self.loss_weights[0]] * len(extra_losses)
            assert len(extra_losses) == len(self.loss_weights), f"{len(extra_losses)}, {len(self.loss_weights)}"
            for p, coef in zip(extra_losses, self.loss_weights):
                if coef != 0 and p is not None:
                    p = coef * p.float() * sample_size
                    loss = loss + p
                    losses.append(p)
        val = (lambda t: t.item()) if sync_logging else (lambda t: t.detach())
        nsent = sample["id"].numel() if "id" in sample else sample["net_input"]["source"].shape[0]
        logging_output = {"loss": val(loss), "ntokens": sample_size, "nsentences": nsent, "sample_size": sample_size}
        for lk in self.log_keys:
            if lk in net_output and lk not in ("logits", "target"):
                v = net_output[lk]
                logging_output[lk] = (float(v) if sync_logging else v) if torch.is_tensor(v) else float(v)
        if len(losses) > 1:
            for i, l in enumerate(losses):
                logging_output[f"loss_{i}"] = val(l)
        if sync_logging:
            logging_output["correct"] = int(out3[1].item()) - int(out3[2].item())
        else:
            logging_output["correct"] = (out3[1] - out3[2]).detach()
        logging_output["count"] = float(sample_size)
        return loss, sample_size, logging_output

    def _fused_outputs(self, net_output, sample, sync_logging):
        """(loss, sample_size, logging_output) from the fused launch: same keys and values as the composed path below."""
        loss, vec = net_output["_fused_loss"].view(()), net_output["_loss_vec"]
        sample_size = net_output["sample_size"]
        host = vec.tolist() if sync_logging else None              # ONE device read for every logged scalar
        val = (lambda i: host[i]) if sync_logging else (lambda i: vec[i])
        nsent = sample["id"].numel() if "id" in sample else sample["net_input"]["source"].shape[0]
        logging_output = {"loss": val(0), "ntokens": sample_size, "nsentences": nsent, "sample_size": sample_size}
        keys = {"prob_perplexity": 5, "code_perplexity": 6, "features_pen": 7}
        for lk in self.log_keys:
            if lk in keys:
                logging_output[lk] = val(keys[lk])
            elif lk in net_output and lk not in ("logits", "target"):
                v = net_output[lk]
                logging_output[lk] = (float(v) if sync_logging else v) if torch.is_tensor(v) else float(v)
        for i in range(3):
            logging_output[f"loss_{i}"] = val(1 + i)
        logging_output["correct"] = int(host[4]) if sync_logging else vec[4]
        logging_output["count"] = float(sample_size)
        return loss, sample_size, logging_output

    @staticmethod
    def reduce_metrics(logging_outputs, metrics=None):
        """fs/criterions/wav2vec_criterion.py:158-212: aggregate the logging outputs of the data-parallel workers / the
        micro-batches of an update.  Same keys, weights and rounding; ``metrics`` is any object with fairseq's
        ``log_scalar`` / ``log_derived`` (``fairseq.metrics`` itself when present) - a private meter set is returned otherwise."""
        m = metrics if metrics is not None else _Meters()
        item = lambda v: float(v.item()) if torch.is_tensor(v) else float(v)      # noqa: E731  (utils.item)
        tot = lambda k: sum(item(log.get(k, 0)) for log in logging_outputs)        # noqa: E731
        loss_sum, ntokens, nsentences, sample_size = tot("loss"), tot("ntokens"), tot("nsentences"), tot("sample_size")
        m.log_scalar("loss", loss_sum / (sample_size or 1) / math.log(2), sample_size, round=3)
        m.log_scalar("ntokens", ntokens)
        m.log_scalar("nsentences", nsentences)
        correct, total = tot("correct"), tot("count")
        m.log_scalar("_correct", correct)
        m.log_scalar("_total", total)
        if total > 0:
            m.log_derived("accuracy", lambda meters: _safe_round(meters["_correct"].sum / meters["_total"].sum, 5)
                          if meters["_total"].sum > 0 else float("nan"))
        builtin = {"loss", "ntokens", "nsentences", "sample_size", "correct", "count"}
        for k in logging_outputs[0]:
            if k not in builtin:
                val = tot(k)
                if k.startswith("loss"):
                    m.log_scalar(k, val / (sample_size or 1) / math.log(2), sample_size, round=3)
                else:
                    m.log_scalar(k, val / len(logging_outputs), round=3)
        return m

    def logging_outputs_can_be_summed(self) -> bool:
        """:215-223 returns ``self.xla``; there is no XLA device here."""
        return False
