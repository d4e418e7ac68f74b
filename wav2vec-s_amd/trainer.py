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


class GradExchange:
    """Data-parallel gradient exchange over a flat fp32 arena (device agnostic: RCCL on MI355X, gloo in
    the CPU tests).  Replaces LegacyDistributedDataParallel.all_reduce_grads
    (fs/distributed/legacy_distributed_data_parallel.py:81-170), which copies every gradient into a
    2^28-element buffer and issues ONE all-reduce strictly after backward.

    Here the backward pass reports milestones ("every element at index >= offset is final"; the arena
    is laid out so that backward finalises it from the end, engine.grad_shapes) and each newly final
    range is all-reduced immediately with async_op=True, in buckets of >= ``bucket_elems`` elements:
    the collective of layer i+1 runs on RCCL's stream while layer i is still being differentiated.
    xGMI is point-to-point (7 links x ~153 GB/s): buckets are kept large (default 16 M floats = 64 MB)
    so the per-collective latency of a ring over 8 GPUs is amortised.  SUM only - the division by the
    global sample_size happens once, inside the fused Adam kernel."""

    def __init__(self, flat: torch.Tensor, dist_module, bucket_elems: int = 16 << 20, group=None, flush_at: int = -1):
        self.flat, self.dist, self.bucket, self.group = flat, dist_module, int(bucket_elems), group
        # flush_at: once a milestone reaches this offset, everything pending goes out even if it is less than a bucket.
        # TrainStep sets it to the start of the encoder's parameters: what is final when only the conv extractor's
        # backward (~2 ms) remains then travels DURING that backward, and the collective left for after the backward is
        # the extractor's 4.2 M elements instead of up to a whole bucket.
        self.flush_at = int(flush_at)
        self.hi = flat.numel()
        self.works = []
        self.launched = []      # (lo, hi) ranges, for tests / tracing

    def begin_step(self):
        self.hi = self.flat.numel()
        self.works, self.launched = [], []

    def on_ready(self, offset: int):
        """Elements [offset, numel) are final.  Launch whole buckets; keep a remainder < bucket for later
        unless offset == 0 (flush everything)."""
        offset = max(0, min(int(offset), self.hi))
        while self.hi - offset >= self.bucket or (offset == 0 and self.hi > 0):
            lo = max(offset, self.hi - self.bucket) if self.hi - offset >= self.bucket else 0
            if offset == 0 and self.hi - lo < self.bucket:
                lo = 0
            self._launch(lo, self.hi)
            self.hi = lo
        if 0 <= offset <= self.flush_at and self.hi > offset:
            self._launch(offset, self.hi)
            self.hi = offset

    def _launch(self, lo, hi):
        if hi <= lo:
            return
        w = self.dist.all_reduce(self.flat[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.works.append(w)
        self.launched.append((lo, hi))

    def finish(self):
        """Flush what is left and make the current stream wait for every collective."""
        self.on_ready(0)
        for w in self.works:
            w.wait()
        self.works = []


class TrainStep:
    """zero_grad -> forward -> criterion -> backward -> [all-reduce] -> Adam.  One call = one micro-batch; every
    ``update_freq``-th call closes an update (fs/trainer.py:632-910 with ``optimization.update_freq``): gradients of
    the micro-batches are SUMMED in the fp32 arena (never zeroed in between), exchanged ONCE - only the last
    micro-batch's backward launches the bucketed all-reduces, as legacy DDP's ``accumulate_grads`` does
    (fs/distributed/legacy_distributed_data_parallel.py:94-99) - and divided by the sample_size summed over micro-batches
    and ranks inside the fused Adam.  update_freq = 1 (default) is the bench's step."""

    def __init__(self, model, criterion, world_size=1, use_optimizer=True, lr=5e-4, betas=(0.9, 0.98), eps=1e-6,
                 weight_decay=0.01, clip_norm=0.0, arena_gib=12.0, update_freq=1):
        self.model, self.criterion, self.world = model, criterion, world_size
        self.flat = FlatParams(model)
        self.use_optimizer = use_optimizer
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip_norm
        self.update_freq = max(1, int(update_freq))
        self.micro = 0            # micro-batches accumulated in the current update
        self.ss_acc = 0
        self.dist = None
        self.exchange = None
        if world_size > 1:
            import torch.distributed as dist
            self.dist = dist
            tail = engine.milestone_offset(self.flat.arena, "post_extract_proj.")
            if tail >= self.flat.arena.numel:
                tail = engine.milestone_offset(self.flat.arena, "encoder.")
            self.exchange = GradExchange(self.flat.arena.flat, dist, flush_at=tail if tail < self.flat.arena.numel else -1)
        self.norm_buf = torch.zeros(1, device=self.flat.p16.device, dtype=torch.float32)
        # all per-step buffers come from one slab (see ops._StepArena); default 12 GiB of the 288 GB
        ops.ARENA.activate(int(arena_gib * (1 << 30)), self.flat.p16.device)

    def __call__(self, sample):
        f = self.flat
        ops.ARENA.reset()
        first, last = self.micro == 0, self.micro == self.update_freq - 1
        if first:
            f.zero_grad()
            self.ss_acc = 0
            if self.exchange is not None:
                self.exchange.begin_step()
        # only the closing micro-batch reports gradient milestones: earlier ones would all-reduce partial sums
        self.model._on_grad_ready = self.exchange.on_ready if (self.exchange is not None and last) else None
        loss, sample_size, log = self.criterion(self.model, sample, sync_logging=False)
        loss.backward()                           # milestones inside launch the bucketed all-reduces
        self.ss_acc += sample_size
        self.micro = 0 if last else self.micro + 1
        if not last:
            return loss.detach()
        total = self.ss_acc
        if self.exchange is not None:
            self.exchange.finish()
            # the global sample_size: mask lengths can differ across ranks (own batches, own masks)
            # torch.full is a fill kernel with the value as a launch argument; torch.tensor([...], device=) would be a
            # synchronous pageable H2D copy on this stream, i.e. the host would wait for the whole backward every step
            ss = torch.full((1,), float(self.ss_acc), device=f.p16.device, dtype=torch.float32)
            self.dist.all_reduce(ss)
            self.ss_dev = ss
            total = None
        if self.use_optimizer:
            f.step += 1
            scale_dev = None
            if total is None:
                scale, scale_dev = 1.0, self.ss_dev.reciprocal_()     # 1 / sum of sample_size over ranks, on device
            else:
                scale = 1.0 / float(total)
            if self.clip > 0:
                self.norm_buf.zero_()
                ops.sumsq(f.arena.flat, self.norm_buf)
                gnorm = float(self.norm_buf.sqrt()) * scale
                if gnorm > self.clip:
                    scale *= self.clip / (gnorm + 1e-6)
            ops.adam_step(f.p32, f.p16, f.m, f.v, f.arena.flat, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1],
                          eps=self.eps, weight_decay=self.wd, step=f.step, scale_host=scale, scale_dev=scale_dev)
        return loss.detach()
