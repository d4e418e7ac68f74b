#!/usr/bin/env python3
"""Records a data-only trajectory of the reference's OWN optimizer pieces, run through
oracle/ref_import.load_optim(): class Adam (fs/optim/adam.py:103-229), clip_grad_norm_ (fs/utils.py:341-386) and
PolynomialDecayLRSchedule (fs/optim/lr_scheduler/polynomial_decay_schedule.py:40-89), sequenced as fs/trainer.py does:

    lr of update 1 = step_update(0)            (_build_optimizer, fs/trainer.py:322-326)
    grads *= 1 / sample_size                   (multiply_grads, :769-774)
    gnorm = clip_grad_norm_(params, clip)      (:781; clip 0 still returns the norm)
    gnorm not finite -> no optimizer.step      (:791-793 raises FloatingPointError; recorded as "skipped": num_updates and
                                                Adam's own step counter do not advance)
    optimizer.step(); num_updates += 1; lr = step_update(num_updates)      (:795, :981-983, :1049-1052)

on an fp32 master of n = 4100 elements in three tensors, yaml betas (0.9, 0.98), eps 1e-6, weight decay 0.01
(wav2vec-S_base_librispeech.yaml:37-44), lr 5e-4 with warm-up 2 and total 6 updates so that warm-up, decay and the floor are
all crossed, 7 gradient sets of which the 4th holds an inf, clip 25 (active on two updates) and clip 0.
tests/golden/optim.npz    (container only; the fixture travels, the reference does not)"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import ref_import  # noqa: E402

R = ref_import.load_optim()
torch.manual_seed(11)
shapes = [(1027,), (48, 64), (1,)]                     # 1027 + 3072 + 1 = 4100 elements
n = sum(int(np.prod(s)) for s in shapes)
p0 = [(torch.randn(s) * 0.1).to(torch.bfloat16).float() for s in shapes]      # a bf16 model's parameters
U = 7
ss = np.array([208.0, 215.0, 199.0, 204.0, 230.0, 210.0, 222.0])             # sample_size = masked frames of the batch
# summed (not yet normalised) gradients; updates 2 and 6 are large enough for clip 25 to bite after the 1/ss division
gscale = [30.0, 8000.0, 25.0, 40.0, 35.0, 12000.0, 20.0]
grads = [[torch.randn(s) * gscale[u] for s in shapes] for u in range(U)]
grads[3][1][5, 7] = float("inf")
betas, eps, wd, lr0, warmup, total = (0.9, 0.98), 1e-6, 0.01, 5e-4, 2, 6.0

out = {"p0": torch.cat([t.reshape(-1) for t in p0]).numpy(), "sample_size": ss, "n": np.int64(n),
       "grads": np.stack([torch.cat([t.reshape(-1) for t in g]).numpy() for g in grads]),
       "hyper": np.array([betas[0], betas[1], eps, wd, lr0, warmup, total])}
for clip in (25.0, 0.0):
    params = [torch.nn.Parameter(t.clone()) for t in p0]
    opt = R.Adam(params, lr=lr0, betas=betas, eps=eps, weight_decay=wd)
    cfg = types.SimpleNamespace(warmup_updates=warmup, total_num_update=total, end_learning_rate=0.0, power=1.0, lr=[lr0],
                                force_anneal=None)
    sched = R.PolynomialDecayLRSchedule(cfg, R.LrHandle(opt))
    sched.step_update(0)
    num_updates = 0
    rec = {k: [] for k in ("lr", "gnorm", "p32", "m", "v", "skipped", "num_updates")}
    for u in range(U):
        for p, g in zip(params, grads[u]):
            p.grad = g.clone()
            p.grad.mul_(1.0 / ss[u])                                 # multiply_grads
        lr_used = opt.param_groups[0]["lr"]
        gnorm = R.clip_grad_norm_(params, clip)
        skipped = not bool(torch.isfinite(gnorm))
        if not skipped:
            opt.step()
            num_updates += 1
            sched.step_update(num_updates)
        flat = lambda key: torch.cat([(opt.state[p][key] if p in opt.state and key in opt.state[p] else torch.zeros_like(p)).reshape(-1)  # noqa: E731
                                      for p in params]).numpy().copy()
        rec["lr"].append(lr_used)
        rec["gnorm"].append(float(gnorm))
        rec["p32"].append(torch.cat([p.detach().reshape(-1) for p in params]).numpy().copy())
        rec["m"].append(flat("exp_avg"))
        rec["v"].append(flat("exp_avg_sq"))
        rec["skipped"].append(skipped)
        rec["num_updates"].append(num_updates)
        print(f"clip {clip:4.1f} update {u + 1}: lr {lr_used:.3e} gnorm {float(gnorm):10.4f} skipped {skipped} num_updates {num_updates}")
    tag = "clip%d" % int(clip)
    for k, v in rec.items():
        out[f"{tag}.{k}"] = np.array(v)
np.savez_compressed(os.path.join(HERE, "optim.npz"), **out)
print("wrote optim.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "optim.npz")) // 1024, "KiB")
