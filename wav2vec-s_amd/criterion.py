"""Mirror of fairseq's ``wav2vec`` criterion (fs/criterions/wav2vec_criterion.py:36-157) for the
InfoNCE configuration wav2vec-S trains with: same constructor arguments, same
``forward(model, sample) -> (loss, sample_size, logging_output)`` contract and logging keys.
The cross entropy and the accuracy counters run in one HIP kernel (w2vs_ce_rows)."""
import torch

from . import ops


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

    def __call__(self, model, sample, reduce=True, sync_logging=True):
        return self.forward(model, sample, reduce, sync_logging)

    def forward(self, model, sample, reduce=True, sync_logging=True):
        """sync_logging=False keeps the logged scalars as device tensors (no .item() host syncs);
        the reference always syncs (wav2vec_criterion.py:110, 128, 133, 151)."""
        net_output = model(**sample["net_input"])
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
