#!/usr/bin/env python3
"""Row f4, the loss head end to end: TransducerOut.train_step (projection d -> V, delay transducer, cross-entropy,
their backward and the two projection-gradient GEMMs) at a speech-translation batch shape on one MI355X.

    python tools/bench_head.py [--B 8 --T 160 --U 48 --d 512 --V 8000 --tokens-per-step 100000]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    for k, v in (("B", 8), ("T", 160), ("U", 48), ("d", 512), ("V", 8000)):
        ap.add_argument("--" + k, type=int, default=v)
    ap.add_argument("--tokens-per-step", type=int, default=100000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "head_bench.json"))
    a = ap.parse_args(argv)
    from wav2vec_s_amd import transducer as tr
    B, T, U, d, V = a.B, a.T, a.U, a.d, a.V
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.randn(B, T, U, d, device="cuda", generator=g)).to(torch.bfloat16).requires_grad_(True)
    proj = torch.nn.Linear(d, V, bias=False).to(torch.bfloat16).cuda()
    xl = torch.tensor(([T, T - 9, T * 3 // 4, T * 5 // 8, T, T // 2, T - 17, T] * B)[:B], device="cuda")
    yl = torch.tensor(([U - 1, U * 5 // 8, U // 4, U - 8, 1, U // 2, U - 1, U * 2 // 3] * B)[:B], device="cuda")
    tg = torch.randint(2, V, (B, U - 1), device="cuda", generator=g)
    head = tr.TransducerOut(proj, delay_scale=1.0, tokens_per_step=a.tokens_per_step)

    def step():
        x.grad = None
        proj.weight.grad = None
        return head.train_step(x * 1.0, tg, xl, yl)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        r = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    cells = B * T * U
    flops = 3 * 2.0 * cells * d * V                      # logits, d x, d W
    rep = {"shape": {"B": B, "T": T, "U": U, "d": d, "V": V, "tokens_per_step": a.tokens_per_step,
                     "micro_batches": -(-B // max(a.tokens_per_step // (T * U), 1))},
           "train_step_ms": round(ms, 3), "cells_per_s": round(cells / (ms / 1e3)),
           "gemm_tflops_equivalent": round(flops / (ms / 1e3) / 1e12, 1), "loss": float(r["loss"])}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
