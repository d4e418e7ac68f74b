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
import contextlib
import ctypes
import os

import torch

from . import _lib, engine, ops

BF16 = torch.bfloat16


class _Roctx:
    """roctx ranges around the phases of an update - the names fairseq gives its own profiler ranges
    (``torch.autograd.profiler.record_function`` in fs/trainer.py:754-795 "reduce-grads" / "multiply-grads" / "clip-grads" /
    "optimizer", fs/tasks/fairseq_task.py:474-478 "forward" / "backward") - so that a rocprofv3 ``--marker-trace`` run can be
    cut by phase.  Push/pop cost ~100 ns when no profiler listens; without the library (or with W2VS_ROCTX=0) they are no-ops."""

    def __init__(self):
        self.lib = None
        if os.environ.get("W2VS_ROCTX", "1") != "0":
            for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    lib = ctypes.CDLL(name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    lib.roctxRangePushA.restype = ctypes.c_int
                    lib.roctxRangePop.argtypes = []
                    lib.roctxRangePop.restype = ctypes.c_int
                    self.lib = lib
                    break
                except (OSError, AttributeError):      # not installed, or a library of that name without the two symbols
                    self.lib = None

    def push(self, name: str):
        if self.lib is not None:
            self.lib.roctxRangePushA(name.encode())

    def pop(self):
        if self.lib is not None:
            self.lib.roctxRangePop()

    @contextlib.contextmanager
    def range(self, name: str):
        self.push(name)
        try:
            yield
        finally:
            self.pop()


ROCTX = _Roctx()
# W2VS_OVERWRITE_WGRADS=0: zero the whole arena at the start of an update and let every weight gradient accumulate (A/B)
OVERWRITE_WGRADS = os.environ.get("W2VS_OVERWRITE_WGRADS", "1") != "0"
WT_CACHE = os.environ.get("W2VS_WT_CACHE", "1") != "0"     # A/B: 0 = transposed weights rebuilt by every backward


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
        # transposed weights of the input-gradient GEMMs, valid while the weights do not change (engine.backward, wt_cache):
        # "have" is cleared by whoever changes a weight - the optimizer step, a load, a hand-written p.data
        self.wt_cache = {"have": set(), "store": {}}
        model._flat = self
        # parameters are VIEWS of p16: load_state_dict / load_pretrained_model / manual init write the bf16 image only, and
        # the next adam_step would rewrite it from a stale fp32 master.  Re-derive the master after every load.
        if hasattr(model, "register_load_state_dict_post_hook"):
            model.register_load_state_dict_post_hook(lambda _m, _inc: self.sync_master_from_model())

    def zero_grad(self, except_ranges=None):
        """Zero the gradient arena; ``except_ranges`` [(offset, numel), ...] are left alone (engine.wgrad_overwrite_ranges:
        the backward WRITES them).  The complement is cleared through one multi-tensor fill; its views are cached per set of
        ranges (LayerDrop makes a dozen different ones)."""
        if not except_ranges:
            self.arena.flat.zero_()
            return
        key = tuple(except_ranges)
        cache = self.__dict__.setdefault("_gap_views", {})
        views = cache.get(key)
        if views is None:
            views, pos = [], 0
            for off, numel in sorted(except_ranges):
                if off > pos:
                    views.append(self.arena.flat[pos:off])
                pos = max(pos, off + numel)
            if pos < self.arena.numel:
                views.append(self.arena.flat[pos:self.arena.numel])
            if len(cache) > 64:
                cache.clear()
            cache[key] = views
        torch._foreach_zero_(views)

    def sync_master_from_model(self):
        """fp32 master <- the bf16 parameters the model currently holds (call after writing ``p.data`` by hand;
        ``load_state_dict`` does it through a hook).  Mirrors FP16Optimizer rebuilding ``fp32_params`` from the model
        (fs/optim/fp16_optimizer.py:53-77)."""
        self.p32.copy_(self.p16)
        self.wt_cache["have"].clear()

    # ---- optimizer state (fs/optim/fp16_optimizer.py:151-177 + torch Adam state: step, exp_avg, exp_avg_sq) ----
    def state_dict(self):
        return {"step": int(self.step), "fp32_params": self.p32.detach().cpu().clone(),
                "exp_avg": self.m.detach().cpu().clone(), "exp_avg_sq": self.v.detach().cpu().clone(),
                "layout": {n: (o, k) for n, (o, k, _) in self.arena.offsets.items()}}

    def load_state_dict(self, sd):
        lay = {n: (o, k) for n, (o, k, _) in self.arena.offsets.items()}
        if sd.get("layout") != lay or sd["fp32_params"].numel() != self.p32.numel():
            raise ValueError("optimizer state was saved for a different parameter layout")
        self.step = int(sd["step"])
        self.p32.copy_(sd["fp32_params"])
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])
        self.p16.copy_(self.p32)          # the working copy is the rounded master, as after every update
        self.wt_cache["have"].clear()


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

    def __init__(self, flat: torch.Tensor, dist_module, bucket_elems: int = 16 << 20, group=None, flush_at: int = -1,
                 wire_dtype: str = "fp32"):
        self.flat, self.dist, self.bucket, self.group = flat, dist_module, int(bucket_elems), group
        self.on_launch = None   # optional callback(lo, hi, work): TrainStep chains the optimizer update of a range behind it
        # flush_at: once a milestone reaches this offset, everything pending goes out even if it is less than a bucket.
        # TrainStep sets it to the start of the encoder's parameters: what is final when only the conv extractor's
        # backward (~2 ms) remains then travels DURING that backward, and the collective left for after the backward is
        # the extractor's 4.2 M elements instead of up to a whole bucket.
        self.flush_at = int(flush_at)
        # wire_dtype "bf16": each range is cast into a bf16 image of the arena (w2vs_f32_to_bf16), THAT is all-reduced -
        # half the bytes on xGMI: 180.6 MB per update for the base model, what the reference moves, which reduces in the
        # model dtype (fs/distributed/legacy_distributed_data_parallel.py:100-115) - and cast back into the fp32 arena once
        # its collective is done.  Local accumulation (update_freq micro-batches, atomics) stays fp32 either way; only the
        # cross-rank sum is rounded.  "fp32" (default) all-reduces the arena in place: exact sums, twice the bytes.
        if wire_dtype not in ("fp32", "bf16"):
            raise ValueError("wire_dtype must be 'fp32' or 'bf16'")
        self.wire_dtype = wire_dtype
        self.wire = torch.empty(flat.numel(), dtype=BF16, device=flat.device) if wire_dtype == "bf16" else None
        self.hi = flat.numel()
        self.works = []
        self.unpack = []        # bf16 wire: (lo, hi, work) whose result is still in the bf16 image
        self.launched = []      # (lo, hi) ranges, for tests / tracing

    def begin_step(self):
        self.hi = self.flat.numel()
        self.works, self.launched, self.unpack = [], [], []

    def on_ready(self, offset: int):
        """Elements [offset, numel) are final.  Launch whole buckets; keep a remainder < bucket for later
        unless offset == 0 (flush everything).

        The launched ranges must be the SAME on every rank whatever milestones it saw (ranks whose host RNG streams differ
        LayerDrop different layers and report different offsets; a rank-dependent boundary would pair collectives of
        different sizes).  So: above ``flush_at`` whole buckets count down from the top of the arena and never reach below
        ``flush_at``; the remainder of that zone goes out as [flush_at, hi) as soon as a milestone reaches flush_at; below it
        only the extractor's milestones remain, which do not depend on LayerDrop, and each flushes what is final."""
        offset = max(0, min(int(offset), self.hi))
        while self.hi > 0:
            floor = self.flush_at if 0 <= self.flush_at < self.hi else 0
            tgt = max(offset, floor)
            if self.hi - tgt >= self.bucket:
                lo = self.hi - self.bucket
            elif offset <= floor and (floor > 0 or offset == 0):
                lo = floor                                 # the zone is complete: its remainder (< bucket) goes out
            else:
                break
            self._launch(lo, self.hi)
            self.hi = lo
        if 0 <= offset <= self.flush_at and offset < self.hi <= self.flush_at:
            self._launch(offset, self.hi)
            self.hi = offset

    def _cast(self, lo, hi, back):
        src, dst = (self.wire, self.flat) if back else (self.flat, self.wire)
        if self.flat.is_cuda:
            (ops.bf16_to_f32 if back else ops.f32_to_bf16)(src[lo:hi], out=dst[lo:hi])
        else:
            dst[lo:hi].copy_(src[lo:hi])          # gloo / CPU tests: a converting copy

    def _launch(self, lo, hi):
        if hi <= lo:
            return
        if self.wire is None:
            w = self.dist.all_reduce(self.flat[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            # ranges whose collective has already finished are unpacked now, beside the backward, not after it
            while self.unpack and self.unpack[0][2].is_completed():
                l0, h0, w0 = self.unpack.pop(0)
                w0.wait()
                self._cast(l0, h0, back=True)
            self._cast(lo, hi, back=False)
            w = self.dist.all_reduce(self.wire[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.unpack.append((lo, hi, w))
        self.works.append(w)
        self.launched.append((lo, hi))
        if self.on_launch is not None:
            self.on_launch(lo, hi, w)

    def finish(self):
        """Flush what is left and make the current stream wait for every collective."""
        self.on_ready(0)
        for w in self.works:
            w.wait()
        for lo, hi, _ in self.unpack:
            self._cast(lo, hi, back=True)
        self.works, self.unpack = [], []


class PolynomialDecayLRSchedule:
    """fs/optim/lr_scheduler/polynomial_decay_schedule.py:40-89 (``lr_scheduler: polynomial_decay`` of both wav2vec-S yamls,
    warm-up 5 000 / 32 000 updates, decay to ``end_learning_rate`` at ``total_num_update`` = max_update).  ``step_update(n)``
    is the rate the trainer applies to the update that FOLLOWS ``n`` completed ones (fs/trainer.py:981-983, 1049-1052):
    the very first update of a run therefore uses 0 (warm-up factor 0 / warmup_updates), as in the reference."""

    def __init__(self, lr, warmup_updates=0, total_num_update=400000, end_learning_rate=0.0, power=1.0):
        assert total_num_update > 0
        self.lr = float(lr[0] if isinstance(lr, (list, tuple)) else lr)
        self.warmup_updates, self.total_num_update = int(warmup_updates), float(total_num_update)
        self.end_learning_rate, self.power = float(end_learning_rate), float(power)
        self.warmup_factor = 1.0 / self.warmup_updates if self.warmup_updates > 0 else 1.0
        self.current = self.warmup_factor * self.lr

    def step_update(self, num_updates):
        if self.warmup_updates > 0 and num_updates <= self.warmup_updates:
            self.warmup_factor = num_updates / float(self.warmup_updates)
            lr = self.warmup_factor * self.lr
        elif num_updates >= self.total_num_update:
            lr = self.end_learning_rate
        else:
            warmup = self.warmup_updates
            pct_remaining = 1 - (num_updates - warmup) / (self.total_num_update - warmup)
            lr = (self.lr - self.end_learning_rate) * pct_remaining ** self.power + self.end_learning_rate
        self.current = lr
        return lr


class TrainStep:
    """zero_grad -> forward -> criterion -> backward -> [all-reduce] -> Adam.  One call = one micro-batch; every
    ``update_freq``-th call closes an update (fs/trainer.py:632-910 with ``optimization.update_freq``): gradients of
    the micro-batches are SUMMED in the fp32 arena (never zeroed in between), exchanged ONCE - only the last
    micro-batch's backward launches the bucketed all-reduces, as legacy DDP's ``accumulate_grads`` does
    (fs/distributed/legacy_distributed_data_parallel.py:94-99) - and divided by the sample_size summed over micro-batches
    and ranks inside the fused Adam.  update_freq = 1 (default) is the bench's step."""

    def __init__(self, model, criterion, world_size=1, use_optimizer=True, lr=5e-4, betas=(0.9, 0.98), eps=1e-6,
                 weight_decay=0.01, clip_norm=0.0, arena_gib=12.0, update_freq=1, lr_scheduler=None, group=None,
                 check_finite=True, wire_dtype="fp32"):
        self.model, self.criterion, self.world = model, criterion, world_size
        self.flat = FlatParams(model)
        self.use_optimizer = use_optimizer
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip_norm
        self.lr_scheduler = lr_scheduler      # e.g. PolynomialDecayLRSchedule; None = constant ``lr``
        self.group = group
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
            self.exchange = GradExchange(self.flat.arena.flat, dist, group=group,
                                         flush_at=tail if tail < self.flat.arena.numel else -1, wire_dtype=wire_dtype)
        dev = self.flat.p16.device
        # The reference computes the gradient norm on EVERY update (clip_grad_norm_ with max_norm 0 still returns the norm,
        # fs/trainer.py:781) and raises FloatingPointError before optimizer.step when it is not finite (:791-793).  Here the
        # norm, the test and the consequence stay on the device: a non-finite norm makes the gradient scale 0, which the fused
        # Adam treats as "skip this update" (nothing is written - master, moments and the bf16 image keep their values), and
        # the flag travels to the host through a pinned buffer that is looked at - without waiting - at the start of later
        # steps, where the error is raised.  check_finite=False (and clip_norm == 0) skips the norm pass (361 MB read).
        self.check_finite = bool(check_finite)
        self.norm_buf = torch.zeros(1, device=dev, dtype=torch.float32)
        self.clip_out = torch.zeros(3, device=dev, dtype=torch.float32)   # [grad scale, gnorm, non-finite flag]
        self._bad_acc = torch.zeros(1, device=dev, dtype=torch.float32)   # sticky: number of skipped (non-finite) updates
        self._bad_reported = 0            # how many of them the host has raised for (and taken out of the step count)
        self._flag_host = torch.zeros(1, dtype=torch.float32).pin_memory() if dev.type == "cuda" else None
        self._flag_event = None
        self._one = None                  # cached d(loss)/d(loss)
        self._before_optimizer = None     # test hook: called right before the norm / Adam launches of an update
        # all per-step buffers come from one slab (see ops._StepArena); default 12 GiB of the 288 GB.  It hands out
        # memory only while a step runs: anything else in the process (validation forward, streaming twin) gets torch's.
        self.arena_bytes = int(arena_gib * (1 << 30))
        ops.ARENA.activate(self.arena_bytes, dev)
        ops.ARENA.suspend()

    def grad_norm(self):
        """Gradient norm of the last update after the 1/sample_size scaling (what fairseq logs as ``gnorm``); available
        when clip_norm > 0 or check_finite.  Reads the device (a sync) - call it when logging, not every step."""
        gn, bad = float(self.clip_out[1]), float(self.clip_out[2])
        if bad:
            raise FloatingPointError("gradients are Nan/Inf")        # fs/trainer.py:791-793
        return gn

    def _report_skipped(self, n_bad):
        """A non-finite update was skipped on the device (nothing written).  The host had already counted it: take it back out
        of Adam's step count (the reference never reaches optimizer.step / set_num_updates for it, fs/trainer.py:791-795), clear
        the sticky device state so that the error is reported ONCE, and raise what the reference raises.  Updates enqueued
        between the skipped one and this report ran with a step count (bias correction, scheduled rate) one too high - call
        ``check()`` after every update when that matters (it is a sync)."""
        # n_bad is the STICKY device count (never zeroed: a skip that happened after the copy the host is looking at would be
        # lost with it); only what has not been reported yet comes out of the step count
        new = int(n_bad) - self._bad_reported
        self._bad_reported = int(n_bad)
        self._flag_event = None
        if new <= 0:
            return
        self.flat.step = max(0, self.flat.step - new)
        self.norm_buf.zero_()
        self.clip_out[2:3].zero_()
        # a caller that catches the error starts a fresh update: no half-accumulated micro-batches, no pending overwrite flag
        self.micro, self.ss_acc = 0, 0
        self.model._wgrad_overwrite = False
        raise FloatingPointError("gradients are Nan/Inf")            # fs/trainer.py:791-793

    def check(self):
        """Synchronous form of the non-finite report: waits for the device, raises FloatingPointError if any update since the
        last report was skipped.  Call it before checkpointing and at the end of training (a non-finite LAST update is otherwise
        never looked at) - or after every update for the reference's exact abort-at-once behaviour."""
        n_bad = float(self._bad_acc[0]) if self.use_optimizer and (self.clip > 0 or self.check_finite) else 0.0
        if n_bad > self._bad_reported:
            self._report_skipped(n_bad)

    finish = check

    def _raise_if_nonfinite(self):
        """Raise for an EARLIER update whose gradient norm was not finite, once its flag has reached the host (no wait)."""
        ev = self._flag_event
        if ev is not None and ev.query():
            self._flag_event = None
            n_bad = float(self._flag_host[0])
            if n_bad > self._bad_reported:
                self._report_skipped(n_bad)    # that update was skipped on the device

    def __call__(self, sample):
        ops.ARENA.activate(self.arena_bytes, self.flat.p16.device)
        try:
            self._raise_if_nonfinite()
            return self._step(sample)
        except BaseException:
            # a step that died between w2vs_sumsq and w2vs_clip_scale_acc would leave its partial sum for the next update
            self.norm_buf.zero_()
            self.micro = 0
            raise
        finally:
            ops.ARENA.suspend()

    def _step(self, sample):
        f = self.flat
        first, last = self.micro == 0, self.micro == self.update_freq - 1
        # The pairwise K split of the grouped weight-gradient launch has workgroups WAIT on their partner: sound only while the
        # whole grid is co-resident, which nobody can promise once RCCL's kernels share the chip with the backward.  An odd
        # (LayerDrop-ped) layer then takes the 256 x 128 single-writer group instead (~10 us per step).  Set per step (the switch
        # is process-wide; another TrainStep without an exchange sets it back).
        _lib.call("w2vs_gemm_tn8_max_split", 1 if self.exchange is not None else 2)
        if first:
            # update_freq > 1: the weights stay as they are until the closing micro-batch's Adam, so their transposes (the B
            # operands of the input-gradient GEMMs: 340 MB of traffic, ~80 us per backward) are made once per update
            f.wt_cache["have"].clear()
            self.model._wt_cache = f.wt_cache if (self.update_freq > 1 and WT_CACHE) else None
            self.ss_acc = 0
            if self.exchange is not None:
                self.exchange.begin_step()
            if not OVERWRITE_WGRADS:
                f.zero_grad()
        # only the closing micro-batch reports gradient milestones: earlier ones would all-reduce partial sums
        self.model._on_grad_ready = self.exchange.on_ready if (self.exchange is not None and last) else None
        with ROCTX.range("forward"):
            loss, sample_size, log = self.criterion(self.model, sample, sync_logging=False)
        if first and OVERWRITE_WGRADS:
            # The arena is cleared AFTER the forward's host phase (which drew LayerDrop) and before the backward: the encoder
            # weight gradients that the grouped single-writer launches produce are WRITTEN by the first micro-batch of an
            # update, so neither their 340 MB fill nor their read inside those kernels happens; everything else - biases,
            # norms, extractor, heads, dropped layers, the pruned last layer - is zeroed as before
            st = getattr(self.model, "_last_state", None)
            ranges = engine.wgrad_overwrite_ranges(st, f.arena) if st is not None else []
            f.zero_grad(ranges)
            self.model._wgrad_overwrite = bool(ranges)
        ss_work = None
        if self.exchange is not None and last:
            # the global sample_size (mask lengths can differ across ranks) is known as soon as the closing micro-batch's
            # forward is: its scalar all-reduce travels under the backward instead of between the last bucket and Adam.
            # torch.full is a fill kernel with the value as a launch argument; torch.tensor([...], device=) would be a
            # synchronous pageable H2D copy on this stream, i.e. the host would wait for the GPU every step
            ss = torch.full((1,), float(self.ss_acc + sample_size), device=f.p16.device, dtype=torch.float32)
            ss_work = self.dist.all_reduce(ss, group=self.group, async_op=True)
            self.ss_dev = ss
        if self._one is None or self._one.device != loss.device:
            self._one = torch.ones((), device=loss.device, dtype=loss.dtype)
        with ROCTX.range("backward"):
            torch.autograd.backward(loss, self._one)  # (a cached seed: loss.backward() fills a fresh one every step); milestones inside launch the bucketed all-reduces
        self.ss_acc += sample_size
        self.micro = 0 if last else self.micro + 1
        if not last:
            return loss.detach()
        total = self.ss_acc
        if self.exchange is not None:
            with ROCTX.range("reduce-grads"):
                self.exchange.finish()
                if ss_work is not None:
                    ss_work.wait()
            total = None
        if self.use_optimizer:
            if self._before_optimizer is not None:
                self._before_optimizer(self)
            lr = self.lr_scheduler.step_update(f.step) if self.lr_scheduler is not None else self.lr
            f.step += 1
            scale_dev = None
            if total is None:
                scale, scale_dev = 1.0, self.ss_dev.reciprocal_()     # 1 / sum of sample_size over ranks, on device
            else:
                scale = 1.0 / float(total)
            if self.clip > 0 or self.check_finite:
                # clip_grad_norm_ (fs/utils.py:341-386) on the gradient AFTER its division by sample_size
                # (fs/trainer.py:769-774): norm, comparison and factor stay on the device; Adam reads the product
                # (0 = non-finite norm = skip the update)
                # norm_buf is zero here: allocated so, and every clip_scale_acc leaves it so.  The same launch adds the
                # non-finite flag to _bad_acc - sticky on the device: a copy that is skipped below loses nothing
                with ROCTX.range("clip-grads"):
                    ops.sumsq(f.arena.flat, self.norm_buf)
                    ops.clip_scale_acc(self.norm_buf, self.clip_out, self._bad_acc, scale_host=scale, scale_dev=scale_dev,
                                       clip=self.clip)
                scale, scale_dev = 1.0, self.clip_out[0:1]
                if self._flag_host is not None and self._flag_event is None:
                    self._flag_host.copy_(self._bad_acc, non_blocking=True)
                    self._flag_event = torch.cuda.Event()
                    self._flag_event.record()
            with ROCTX.range("optimizer"):
                ops.adam_step(f.p32, f.p16, f.m, f.v, f.arena.flat, lr=lr, beta1=self.betas[0], beta2=self.betas[1],
                              eps=self.eps, weight_decay=self.wd, step=f.step, scale_host=scale, scale_dev=scale_dev)
            self.last_lr = lr
            f.wt_cache["have"].clear()            # the weights have changed
            self.model._wt_cache = None
        return loss.detach()
