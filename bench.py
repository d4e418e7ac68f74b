#!/usr/bin/env python3
"""Headline benchmark: wav2vec-S base pre-training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch: conv feature extractor ->
block-causal encoder -> Gumbel quantizer -> InfoNCE + diversity + feature-penalty loss ->
backward of all of it (+ gradient all-reduce over RCCL when N > 1) + fused Adam update.
Workload = BASELINE.json configs[1]: base model (12 x 768), bf16, batch 8 x 175 000 samples
(1.4 M samples = 87.5 audio-seconds) per GPU, yaml dropouts / LayerDrop / sampled contexts on.
Weak scaling: every rank gets its own batch.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The host thread only issues launches; idle OpenMP workers spinning after torch's small CPU ops
# (mask / negative sampling) would burn the container's CPU quota and get the launch thread
# throttled for tens of ms.  Keep the pools small and passive; the cpu_baseline leg raises the
# thread count explicitly for its own run.
os.environ.setdefault("OMP_NUM_THREADS", "4")
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
os.environ.setdefault("MKL_NUM_THREADS", "4")

import numpy as np  # noqa: E402
import torch  # noqa: E402

B_PER_GPU, L_SAMPLES, SR = 8, 175000, 16000
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def cpu_baseline(threads, large=False):
    """The oracle (a CPU port of the reference algorithm, fp32) timed on the host cores over a bounded
    sample of the same workload: base model, cfgA = 2 x 160 000 samples (20 audio-seconds; BASELINE configs[0], the
    reference's own CPU-runnable case), fwd+loss+bwd.  The restatement / reference time ratio
    measured in the build container is recorded in BASELINE.md (tools/cpu_ratio.py)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import w2vs_oracle as O
    torch.set_num_threads(threads)
    cfg = O.OracleCfg()
    Bc, Lc = 2, 160000
    if large:   # wav2vec-S_large_librivox.yaml:52-70; bounded sample: one 10 s utterance
        cfg = O.OracleCfg(encoder_layers=24, encoder_embed_dim=1024, encoder_ffn_embed_dim=4096, encoder_attention_heads=16,
                          layer_norm_first=True, conv_bias=True, feature_grad_mult=1.0, final_dim=768, loss_weights=(0.1, 0.0))
        Bc, Lc = 1, 160000
    P = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=1).items()}
    g = torch.Generator().manual_seed(1234)
    src = torch.randn(Bc, Lc, generator=g)
    T = O.conv_out_lengths(Lc, cfg.conv_layers)[-1]
    np.random.seed(1234)
    mask = torch.from_numpy(O.compute_mask_indices((Bc, T), None, 0.65, 10, "static", 0, min_masks=2))
    M = int(mask[0].sum())
    torch.manual_seed(1234)
    neg = O.sample_negative_indices(Bc, M, 100)
    noise = -torch.empty(Bc * M * 2, 320).exponential_().log()
    times = []
    for it in range(3):
        for p in P.values():
            p.grad = None
        t0 = time.perf_counter()
        out = O.forward_loss(P, src, cfg, mask_indices=mask, neg_idx=neg, main_context=16, right_context=8, tau=2.0,
                             gumbel_noise=noise)
        out["loss"].backward()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    return {"value": round(Bc * Lc / SR / t, 3), "unit": "audio-s/s", "cores": threads, "kind": "port",
            "sample": "oracle fp32, %s model, %d x %d samples, fwd+loss+bwd, median of 2 after 1 warm-up, %.2f s/step" % (
                "large" if large else "base", Bc, Lc, t)}


def _cpu_baseline_rnnt(acts1, labels1, T, U, gpu_cost):
    """cpu_baseline leg of --workload rnnt: warp_transducer's own CpuRNNT (compiled into oracle/_ref by `make -C oracle ref`)
    on ONE utterance; it takes log-probabilities, so numpy's log-softmax of that utterance is timed with it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rnnt_oracle as R
    if not R.RefCpuRnnt.available():
        return "unavailable: oracle/_ref/libwarprnnt_cpu.so not built"
    ref = R.RefCpuRnnt()
    x = acts1.astype(np.float64)
    t0 = time.perf_counter()
    lp = x + R.log_softmax_denom(x)[..., None]
    c, _ = ref.loss_and_logprob_grads(lp, labels1, [T], [U - 1])
    dt = time.perf_counter() - t0
    return {"kind": "reference", "cores": 1, "value": round(T * U / dt), "unit": "lattice cells/s", "seconds": round(dt, 3),
            "sample": "utterance 0 (%d x %d cells, V=%d), numpy log-softmax + warp_transducer CpuRNNT (plain RNN-T, no delay "
                      "terms)" % (T, U, acts1.shape[-1]),
            "cost_equal": bool(abs(float(c[0]) - gpu_cost) < 2e-4 * abs(float(c[0])))}


def _cpu_baseline_data(idx, sizes, product_batches):
    """cpu_baseline leg of --workload data: the reference's Cython batch_by_size_vec compiled into oracle/_ref."""
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(ref_dir):
        return {"reference_cython": "unavailable: oracle/_ref not built"}
    sys.path.insert(0, ref_dir)
    try:
        import data_utils_fast as fast          # the compiled reference; /root/reference is not needed at run time
    except ImportError as e:
        return {"reference_cython": "unavailable: %r" % (e,)}
    t0 = time.perf_counter()
    for _ in range(5):
        rb = fast.batch_by_size_vec(idx, sizes, 1400000, -1, 8)
    tr = (time.perf_counter() - t0) / 5
    same = len(rb) == len(product_batches) and all(np.array_equal(x, y) for x, y in zip(rb, product_batches))
    return {"reference_cython_ms": round(tr * 1e3, 3), "identical_to_reference": bool(same)}


def side_workload(name, no_cpu):
    """Rows f1 / f3 / f4: one JSON line each, same spirit as the headline line (tools/bench_*.py do the GPU part)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    if name == "stream":
        import bench_stream
        return bench_stream.main([])
    if name == "data":
        import bench_data
        return bench_data.main([], cpu_baseline=None if no_cpu else _cpu_baseline_data)
    if name == "caat":
        import bench_caat
        return bench_caat.main([])
    import bench_rnnt
    return bench_rnnt.main([], cpu_baseline=None if no_cpu else _cpu_baseline_rnnt)


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks the way the driver's own line does
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`) as a
    CHILD process - this process has not touched the GPU (no HIP call so far; counting devices does not initialise it) and
    never will - relay its output and leave with its exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("W2VS_REHEARSE_ONE_GPU") != "1":
        print("bench.py: --gpus %d but %d GPUs are visible; refusing to time fewer ranks than asked for" % (n, have), file=sys.stderr)
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith("{") and '"metric"' in ln]
    for ln in lines:
        if ln not in js:
            print(ln, file=sys.stderr)              # launcher chatter stays off stdout: ONE JSON line
    if proc.returncode != 0 or len(js) != 1:
        print("bench.py: the %d-rank launch failed (exit code %d, %d result lines)" % (n, proc.returncode, len(js)), file=sys.stderr)
        sys.exit(proc.returncode or 3)
    rec = json.loads(js[0])
    if rec.get("n_gpus") != n or rec.get("ranks_seen") != n:
        print("bench.py: asked for %d ranks, the run reports n_gpus=%r ranks_seen=%r" % (n, rec.get("n_gpus"), rec.get("ranks_seen")),
              file=sys.stderr)
        sys.exit(3)
    print(js[0])
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the constant-context / LayerDrop-0 variant run")
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--samples", type=int, default=L_SAMPLES)
    ap.add_argument("--update-freq", type=int, default=1,
                    help="micro-batches per optimizer update (BASELINE configs[2] quotes 8); a timed 'step' stays ONE micro-batch "
                         "pass, so K steps = K micro-batches and K / update_freq updates + gradient exchanges")
    ap.add_argument("--wire", default=os.environ.get("W2VS_WIRE", "bf16"), choices=["fp32", "bf16"],
                    help="N > 1: dtype the gradient all-reduce moves.  bf16 (default) = the reference's model-dtype exchange, 180.6 MB "
                         "per update (SURVEY.md section 8e); fp32 = the arena in place, exact sums, twice the xGMI bytes")
    ap.add_argument("--no-gemm-peak", action="store_true",
                    help="skip the 8192^3 calibration GEMM behind the timed region (profiler passes: its launches would be "
                         "pooled with the step's own launches of the same kernel symbol)")
    ap.add_argument("--length-mix", action="store_true",
                    help="--workload large only: Libri-light-shaped batches - every step draws its utterance lengths from "
                         "U[160 000, 320 000] and crops to the batch minimum (raw_audio_dataset.py:131-151, pad=False)")
    ap.add_argument("--workload", default="pretrain", choices=["pretrain", "stream", "data", "rnnt", "caat", "large"],
                    help="pretrain (default) = the headline step; stream / data / rnnt = the SURVEY section 8 rows f1 / f3 / f4 "
                         "measurements (tools/bench_*.py) with their CPU baselines attached here")
    args = ap.parse_args()
    if args.workload not in ("pretrain", "large"):
        return side_workload(args.workload, args.no_cpu_baseline)
    large = args.workload == "large"
    if large:        # BASELINE configs[3]: 3 x 320 000 samples per GPU (max_tokens 1.2 M, wav2vec-S_large_librivox.yaml:16-22)
        if args.batch == B_PER_GPU:
            args.batch = 3
        if args.samples == L_SAMPLES:
            args.samples = 320000

    force_dist = os.environ.get("W2VS_FORCE_DIST") == "1"   # exercise the RCCL path with a 1-rank group
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not force_dist:
        return self_launch(args.gpus)             # `python bench.py --gpus N`: start the N ranks here, before any GPU call
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not force_dist:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # W2VS_REHEARSE_ONE_GPU=1: the driver's multi-rank launch line on a ONE-GPU box - every rank on cuda:0, gloo instead of
        # RCCL (which wants a GPU per rank).  Checks the rendezvous, the exchange from real backward milestones, the barriers
        # and the one JSON line; its timing means nothing (two ranks share the card, the arena travels through host memory)
        rehearse = os.environ.get("W2VS_REHEARSE_ONE_GPU") == "1"
        if rehearse:
            local_rank = 0
        elif world > torch.cuda.device_count():
            print("bench.py: %d ranks but only %d GPUs are visible (one process per GPU)" % (world, torch.cuda.device_count()),
                  file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        backend = "gloo" if rehearse else "nccl"
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        # the ranks the collective library itself sees: one all-reduce of ones over the group that carries the gradients
        seen = torch.ones(1, device=torch.device("cuda", local_rank))
        dist.all_reduce(seen)
        ranks_seen = int(seen.item())
        if ranks_seen != (1 if force_dist and world == 1 else args.gpus) or dist.get_world_size() != ranks_seen:
            print("bench.py: --gpus %d but %s saw %d ranks (world size %d)" % (args.gpus, backend, ranks_seen, dist.get_world_size()),
                  file=sys.stderr)
            sys.exit(2)
    else:
        ranks_seen = 1
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import wav2vec_s_amd as w
    from wav2vec_s_amd import ops, trainer
    cfg = w.large_librivox_config() if large else w.base_librispeech_config()
    torch.manual_seed(1)                       # identical replicas on every rank
    model = w.Wav2VecSModel(cfg).to(torch.bfloat16).to(dev).train()
    crit = w.Wav2vecCriterion(infonce=True, loss_weights=[0.1, 0.0] if large else [0.1, 10.0],
                              log_keys=["prob_perplexity", "code_perplexity", "temp"])
    # optimizer section of the yamls: adam (0.9, 0.98), eps 1e-6, wd 0.01, polynomial decay from 5e-4 with 5 000 / 32 000 warm-up
    # updates to max_update 400 000; the large yaml clips the gradient norm at 25 (wav2vec-S_large_librivox.yaml:38-50)
    sched = trainer.PolynomialDecayLRSchedule(5e-4, warmup_updates=32000 if large else 5000, total_num_update=400000)
    sched.step_update(100000)                  # a mid-training rate for the synthetic run (update 0 would apply lr = 0)
    step_fn = trainer.TrainStep(model, crit, world_size=max(world, 2) if force_dist else world,
                                use_optimizer=not args.no_optimizer, update_freq=args.update_freq,
                                lr=sched.current, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01,
                                clip_norm=25.0 if large else 0.0, arena_gib=40.0 if large else 12.0,
                                check_finite=os.environ.get("W2VS_CHECK_FINITE", "1") == "1", wire_dtype=args.wire)
    B, L = args.batch, args.samples
    g = torch.Generator().manual_seed(1234 + rank)
    source = torch.randn(B, L, generator=g)
    if large:                                   # task.normalize: true (raw_audio_dataset.py:69-72)
        source = torch.nn.functional.layer_norm(source, (L,))
    source = source.to(torch.bfloat16).to(dev)
    # host RNG streams are seeded IDENTICALLY on every rank, as the reference does (fairseq_cli/train.py:67-68:
    # np.random.seed(seed); utils.set_torch_seed(seed)): same mask lengths, same LayerDrop decisions and same sampled
    # block contexts everywhere - ranks differ in their audio only, so no rank waits for one that dropped fewer layers
    np.random.seed(1234)
    random.seed(1234)
    torch.manual_seed(1234)
    torch.cuda.manual_seed(1234)
    sample = {"net_input": {"source": source}}
    mix_rng = np.random.RandomState(4321)       # its own stream: the host draws of the model keep theirs
    audio_done = [0.0]

    def next_sample():
        """The fixed batch, or (--length-mix) a Libri-light-shaped one: B lengths from U[160 000, 320 000] cropped to their
        minimum, as the reference's collater does with pad=False (fs/data/audio/raw_audio_dataset.py:131-151)."""
        if not (large and args.length_mix):
            audio_done[0] += B * L / SR
            return sample
        lmin = int(mix_rng.randint(160000, 320001, size=B).min())
        audio_done[0] += B * lmin / SR
        return {"net_input": {"source": source[:, :lmin].contiguous()}}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from wav2vec_s_amd import flops as flops_mod

    def timed(n_warm, n_steps, first_update):
        """W untimed + K timed steps bracketed by barrier + synchronize; returns (seconds, algorithmic FLOPs of the K steps
        as actually drawn - sampled contexts and LayerDrop change them -, last loss)."""
        for i in range(n_warm):
            model.set_num_updates(first_update + i)
            step_fn(next_sample())
        barrier()
        fl = 0.0
        fl_exec[0] = 0.0
        audio_done[0] = 0.0
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]   # per-step GPU time without a host sync
        marks[0].record()
        t0 = time.perf_counter()
        for i in range(n_steps):
            model.set_num_updates(first_update + n_warm + i)
            loss = step_fn(next_sample())
            marks[i + 1].record()
            fl += flops_mod.step_flops_from_state(model._last_state)     # host arithmetic on the step's own draws
            fl_exec[0] += flops_mod.executed_step_flops_from_state(model._last_state)
        barrier()
        el = time.perf_counter() - t0
        per_step_ms[:] = [marks[i].elapsed_time(marks[i + 1]) for i in range(n_steps)]
        return el, fl, loss

    per_step_ms = []
    fl_exec = [0.0]

    for i in range(args.warmup):
        model.set_num_updates(i)
        step_fn(sample)
    barrier()
    ops.GEMM_TIMER.enable()
    elapsed, step_flops, loss_t = timed(0, args.steps, args.warmup)
    exec_flops = fl_exec[0]
    ops.GEMM_TIMER.disable()
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    ms_median = float(np.median(per_step_ms)) if per_step_ms else None
    ms_min_max = [round(min(per_step_ms), 3), round(max(per_step_ms), 3)] if per_step_ms else None
    audio_s = audio_done[0] / args.steps        # per step and GPU (== B * L / SR unless --length-mix)
    value = world * audio_s * args.steps / elapsed
    roof = ops.GEMM_TIMER.report(PEAK_BF16_TFLOPS)
    # step level: algorithmic FLOPs of the steps actually run / their time / the bf16 MFMA peak
    ach = step_flops / elapsed / 1e12
    # frac: the reference's arithmetic (SURVEY 8d calculator) over the step time; frac_executed: the same net of the rows of
    # the last kept layer this build never computes (flops.pruned_forward_flops) - the MFMA work actually issued
    ach_x = exec_flops / elapsed / 1e12
    roof["step"] = {"flops_per_step": round(step_flops / args.steps / 1e12, 4), "unit": "TFLOP", "achieved_tflops": round(ach, 1),
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                    "flops_executed_per_step": round(exec_flops / args.steps / 1e12, 4), "executed_tflops": round(ach_x, 1),
                    "frac_executed": round(ach_x / PEAK_BF16_TFLOPS, 4),
                    "note": "this run's own draws (sampled contexts%s)" % ("" if large else ", LayerDrop 0.05")}
    st_main = model._last_state
    # HBM-side bytes per launch of the dominant kernel come from SEPARATE rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    # of this same command (counters cannot be read in-process).  The newest committed summary is quoted only if it was
    # collected from THIS build of the kernels (it records the sha256 of csrc/gemm.hip); otherwise traffic stays null.
    try:
        import glob
        import hashlib
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_hbm_traffic.json")))
        pmc = json.load(open(cands[-1]))
        src_hash = hashlib.sha256(open(os.path.join(ROOT, "wav2vec-s_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:16]
        key = {"gemm_tn_lc_kernel": "w2vs::gemm_tn_lc_kernel<true>", "gemm_tn_kernel": "w2vs::gemm_tn_kernel",
               "gemm_tn_group_kernel": "w2vs::gemm_tn_group_kernel",
               "gemm_tn8_group_kernel": "w2vs::gemm_tn8_group_kernel"}.get(roof["kernel"].split(" ")[0], roof["kernel"].split("<")[0])
        if pmc.get("_gemm_hip_sha256") != src_hash:
            roof["traffic_note"] = "%s was collected from another build of gemm.hip: not quoted" % os.path.basename(cands[-1])
        elif key in pmc:
            roof["traffic"] = int((pmc[key]["read_MB_per_launch_corrected"] + pmc[key]["write_MB_per_launch"]) * 1e6)
            roof["traffic_note"] = ("bytes/launch, rocprofv3 --pmc (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), profiles/"
                                    + os.path.basename(cands[-1]))
    except Exception:
        pass
    # "a measured large-GEMM peak on the box" beside the vendor peak - one 8192^3 bf16 NT GEMM of the
    # product's own kernel (automatic choice = the 8-phase 256 x 256 kernel), outside the timed region
    if rank == 0 and not args.no_gemm_peak:
        try:
            gx = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
            gw = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
            for _ in range(2):
                ops.linear_fwd(gx, gw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.linear_fwd(gx, gw)
            e1.record()
            torch.cuda.synchronize()
            roof["measured_gemm_peak_tflops"] = round(5 * 2.0 * 8192 ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
            roof["measured_gemm_peak_note"] = "w2vs_gemm_nt, 8192^3 bf16, random operands, same process, after the timed region"
            roof["step"]["frac_of_measured_peak"] = round(ach / roof["measured_gemm_peak_tflops"], 4)
            roof["step"]["frac_executed_of_measured_peak"] = round(ach_x / roof["measured_gemm_peak_tflops"], 4)
            del gx, gw
        except Exception as e:          # noqa: BLE001
            roof["measured_gemm_peak_note"] = "not measured: %r" % (e,)
    variants = {}
    if world == 1 and not args.no_variants:
        # also the constant (16, 8) context with LayerDrop off - the step-level roofline number
        model.cfg.context_type, model.cfg.encoder_layerdrop = "constant", 0.0
        el2, fl2, _ = timed(3, args.steps, args.warmup + args.steps)
        a2 = fl2 / el2 / 1e12
        variants["constant_context_layerdrop_0"] = {
            "ms_per_step": round(el2 / args.steps * 1e3, 3), "value": round(audio_done[0] / el2, 2), "unit": "audio-s/s",
            "flops_per_step": round(fl2 / args.steps / 1e12, 4), "achieved_tflops": round(a2, 1),
            "frac_of_bf16_peak": round(a2 / PEAK_BF16_TFLOPS, 4)}
        roof["step_constant_context_layerdrop_0"] = {"flops_per_step": round(fl2 / args.steps / 1e12, 4),
                                                     "achieved_tflops": round(a2, 1), "frac": round(a2 / PEAK_BF16_TFLOPS, 4)}
    st = st_main
    out = {
        "metric": "audio-seconds/s/GPU, wav2vec-S %s pretrain step, 1/2/4/8 MI355X" % ("large" if large else "base"),
        "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "ms_per_step_median": None if ms_median is None else round(ms_median, 3),
        "ms_per_step_min_max": ms_min_max,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": ("wav2vec-S large (24L d1024 pre-LN, 315M params, BASELINE configs[3]) pretrain step: fwd + InfoNCE/"
                                "diversity loss + bwd%s%s (clip_norm 25); %d x %d samples (%.1f audio-s) per GPU; yaml dropouts, "
                                "sampled block contexts; random-init weights; unread rows of the last encoder layer pruned (exact)"
                                if large else
                                "wav2vec-S base (12L d768, 90.3M params) pretrain step: fwd + InfoNCE/diversity/penalty loss + "
                                "bwd%s%s; %d x %d samples (%.1f audio-s) per GPU; yaml dropouts, LayerDrop 0.05, sampled "
                                "block contexts; random-init weights; unread rows of the last encoder layer pruned (exact)") % (
                                   (" + %s grad all-reduce (%s on the wire)" % ("RCCL" if backend == "nccl" else "gloo (one-GPU rehearsal, not RCCL)", args.wire)) if world > 1 else "",
                                   ("" if args.no_optimizer else " + fused Adam") + (
                                       "" if args.update_freq == 1 else " (update_freq %d: exchange + Adam every %d-th step)" % (
                                           args.update_freq, args.update_freq)), B, L, audio_s),
                   "length_mix": ("Libri-light shape: per step B lengths from U[160 000, 320 000] cropped to their minimum "
                                  "(raw_audio_dataset.py:131-151); %.1f audio-s per step on average" % audio_s) if (large and args.length_mix) else None,
                   "global_batch_samples": world * B * L, "per_gpu_audio_s_per_s": round(value / world, 2),
                   "last_loss_per_sample": round(float(loss_t) / max(1, st.B * st.M), 4),
                   "tokens_last_step": {"T": st.T, "N": st.N, "M": st.M, "m": st.m, "r": st.r}},
        "roofline": roof,
    }
    if variants:
        out["variants"] = variants
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the CPU leg runs at N = 1 only
        threads = min(os.cpu_count() or 1, 16)
        out["cpu_baseline"] = cpu_baseline(threads, large)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
