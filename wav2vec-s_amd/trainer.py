"""The micro-step around the model: flat parameter / gradient storage, fused Adam, and the
data-parallel gradient exchange (e, f2).

Reference behaviour restated (fs/trainer.py:632-910, fs/distributed/legacy_distributed_data_parallel.py:
81-170, fs/optim/fp16_optimizer.py, fs/optim/adam.py): gradients of all micro-batches are summed,
all-reduced (sum) across ranks, divided by the global sample_size, optionally clipped, and applied by
Adam to an fp32 master copy whose bf16 image the model computes with.

MI355X design: ONE flat bf16 parameter buffer, ONE flat fp32 master/m/v, ONE flat fp32 gradient
arena that the backward kernels accumulate into directly - the all-reduce and the optimizer each
touch a single contiguous range, no per-tensor launches, no gradient copies.
"""
import torch

from . import engine, ops

BF16 = torch.bfloat16


class FlatParams:
    """Re-homes every parameter of the model into one flat bf16 buffer (conv weights stored
    tap-major [Cout, k, Cin], exposed to ``state_dict`` as permuted views, so checkpoints keep the
    reference layout)."""

    def __init__(self, model):
        named = list(model.named_parameters())
        dev = named[0][1].device
        W = {n: p for n, p in named}
        shapes = engine.grad_shapes(model.cfg, W)
        self.arena = engine.Arena(shapes, dev)
        n_tot = self.arena.numel
        self.p16 = torch.zeros(n_tot, device=dev, dtype=BF16)
        self.packed = {}
        with torch.no_grad():
            for n, p in named:
                off, numel, shp = self.arena.offsets[n]
                view = self.p16[off:off + numel].view(shp)
                if n.startswith("feature_extractor.conv_layers.") and n.endswith(".0.weight"):
                    view.copy_(p.detach().to(BF16).permute(0, 2, 1))
                    self.packed[n] = view.view(shp[0], -1)
                    p.data = view.permute(0, 2, 1)
                else:
                    view.copy_(p.detach().to(BF16))
                    p.data = view
        self.p32 = self.p16.float()
        self.m = torch.zeros_like(self.p32)
        self.v = torch.zeros_like(self.p32)
        self.step = 0
        model._flat = self

    def zero_grad(self):
        self.arena.flat.zero_()


class TrainStep:
    """zero_grad -> forward -> criterion -> backward -> [all-reduce] -> Adam.  One call = one update
    with update_freq = 1 (the bench's step)."""

    def __init__(self, model, criterion, world_size=1, use_optimizer=True, lr=5e-4, betas=(0.9, 0.98), eps=1e-6,
                 weight_decay=0.01, clip_norm=0.0, arena_gib=12.0):
        self.model, self.criterion, self.world = model, criterion, world_size
        self.flat = FlatParams(model)
        self.use_optimizer = use_optimizer
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip_norm
        self.dist = None
        if world_size > 1:
            import torch.distributed as dist
            self.dist = dist
        self.norm_buf = torch.zeros(1, device=self.flat.p16.device, dtype=torch.float32)
        # all per-step buffers come from one slab (see ops._StepArena); default 12 GiB of the 288 GB
        ops.ARENA.activate(int(arena_gib * (1 << 30)), self.flat.p16.device)

    def __call__(self, sample):
        f = self.flat
        ops.ARENA.reset()
        f.zero_grad()
        loss, sample_size, log = self.criterion(self.model, sample, sync_logging=False)
        loss.backward()
        total = sample_size
        if self.dist is not None:
            # sum of gradients over ranks (legacy_ddp: pre-divide by world, trainer multiplies back)
            self.dist.all_reduce(f.arena.flat, op=self.dist.ReduceOp.SUM)
            total = sample_size * self.world      # every rank has the same M in this workload
        if self.use_optimizer:
            f.step += 1
            scale = 1.0 / float(total)
            if self.clip > 0:
                self.norm_buf.zero_()
                ops.sumsq(f.arena.flat, self.norm_buf)
                gnorm = float(self.norm_buf.sqrt()) * scale
                if gnorm > self.clip:
                    scale *= self.clip / (gnorm + 1e-6)
            ops.adam_step(f.p32, f.p16, f.m, f.v, f.arena.flat, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1],
                          eps=self.eps, weight_decay=self.wd, step=f.step, scale_host=scale)
        return loss.detach()
