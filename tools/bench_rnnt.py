#!/usr/bin/env python3
"""Row f4 measurement: the delay-transducer loss at a speech-translation batch shape on one MI355X.

    python tools/bench_rnnt.py [--B 8 --T 160 --U 48 --V 8000] [--out gpurun_out/rnnt_bench.json]

Per kernel (HIP events on the launching stream, inputs resident): rows (log-softmax denominators + blank/label
extraction), lattice (alpha/beta + delay recursions), grad (gradient rows).  HBM roofline: algorithmic bytes =
4 V per valid row read (rows), 4 V read + 4 V written per valid row and 4 V written per padded row (grad).
The CPU baseline (warp_transducer's own CPU RNN-T compiled into oracle/_ref, one utterance of the batch) is timed only
through `python bench.py --workload rnnt`: bench.py's cpu_baseline leg owns every use of oracle/.
"""
import argparse
import json
import os
import sys
import numpy as np  # noqa: F401
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ev_time(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


def main(argv=None, cpu_baseline=None):
    ap = argparse.ArgumentParser()
    for k, v in (("B", 8), ("T", 160), ("U", 48), ("V", 8000)):
        ap.add_argument("--" + k, type=int, default=v)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rnnt_bench.json"))
    a = ap.parse_args(argv)
    from wav2vec_s_amd import transducer as tr
    lib = tr._rnnt_lib()
    B, T, U, V = a.B, a.T, a.U, a.V
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.randn(B, T, U, V, device="cuda", generator=g) * 2
    xl = torch.tensor(([T, T - 9, T * 3 // 4, T * 5 // 8, T, T // 2, T - 17, T] * B)[:B], dtype=torch.int32, device="cuda")
    yl = torch.tensor(([U - 1, U * 5 // 8, U // 4, U - 8, 1, U // 2, U - 1, U * 2 // 3] * B)[:B], dtype=torch.int32, device="cuda")
    lab = torch.randint(1, V, (B, U - 1), device="cuda", dtype=torch.int32, generator=g)
    dv = tr.delay_cost_zero(acts, xl, yl)
    ws = torch.empty(tr.workspace_bytes(T, U, B, True) // 4, dtype=torch.float32, device="cuda")
    costs = torch.empty(3, B, dtype=torch.float32, device="cuda")
    grads = torch.empty_like(acts)
    opt = tr._options(acts, 0)
    args_f = (acts.data_ptr(), lab.data_ptr(), yl.data_ptr(), xl.data_ptr(), dv.data_ptr(), V, B, costs.data_ptr(),
              ws.data_ptr(), 1.0, opt, None, 0)
    args_b = (acts.data_ptr(), grads.data_ptr(), lab.data_ptr(), yl.data_ptr(), xl.data_ptr(), dv.data_ptr(), V, B,
              ws.data_ptr(), 1.0, 1.0, 0, None, 0, 1.0, opt, None, 0)
    t_f = ev_time(lambda: lib.w2vs_rnnt_forward_async(*args_f))
    t_b = ev_time(lambda: lib.w2vs_rnnt_backward_async(*args_b))
    valid = int((xl.long() * (yl.long() + 1)).sum())
    rows = B * T * U
    bytes_rows = 4 * V * valid
    bytes_grad = 8 * V * valid + 4 * V * (rows - valid)
    # the lattice alone: a forward call on a V = 8 problem of the same lattice shape has negligible row work
    acts_s = acts[..., :8].contiguous()
    lab_s = (lab % 7 + 1).to(torch.int32)
    args_s = (acts_s.data_ptr(), lab_s.data_ptr(), yl.data_ptr(), xl.data_ptr(), dv.data_ptr(), 8, B, costs.data_ptr(),
              ws.data_ptr(), 1.0, opt, None, 0)
    t_lat = ev_time(lambda: lib.w2vs_rnnt_forward_async(*args_s))
    lib.w2vs_rnnt_forward_async(*args_f)
    rep = {"shape": {"B": B, "T": T, "U": U, "V": V, "valid_cells": valid, "cells": rows},
           "forward_us": round(t_f, 1), "backward_us": round(t_b, 1), "lattice_us_approx": round(t_lat, 1),
           "rows": {"algorithmic_MB": round(bytes_rows / 1e6, 1), "us": round(t_f - t_lat, 1),
                    "achieved_GBps": round(bytes_rows / max(t_f - t_lat, 1e-3) / 1e3, 1)},
           "grad": {"algorithmic_MB": round(bytes_grad / 1e6, 1), "us": round(t_b, 1),
                    "achieved_GBps": round(bytes_grad / t_b / 1e3, 1)},
           "hbm_peak_GBps": 8000,
           "loss_and_grad_ms": round((t_f + t_b) / 1e3, 3),
           "valid_cells_per_s": round(valid / ((t_f + t_b) / 1e6))}
    if cpu_baseline is not None:                    # bench.py's cpu_baseline leg hands in the checker-side callable
        b = 0
        Tb, Ub = int(xl[b]), int(yl[b]) + 1
        rep["cpu_baseline"] = cpu_baseline(acts[b:b + 1, :Tb, :Ub].cpu().numpy(), lab[b:b + 1, :Ub - 1].cpu().numpy(), Tb, Ub,
                                           float(costs[0, b]))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
