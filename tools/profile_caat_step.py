#!/usr/bin/env python3
"""BASELINE configs[4] step (tools/caat_shapes.py): host issue time vs total, per stage.
    python tools/profile_caat_step.py [script|arch]          (W2VS_CPROFILE=1: + cProfile of the issuing thread)"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import caat_shapes  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "script"
random.seed(1234)
built = caat_shapes.build(shape)
if shape == "script":
    built["jn"].step_mode = "constant"       # one group size for every profiled call (the script's own 64)


def stage_times(sync):
    t = [time.perf_counter()]

    def mark():
        if sync:
            torch.cuda.synchronize()
        t.append(time.perf_counter())
    built["step"](mark)
    torch.cuda.synchronize()
    t.append(time.perf_counter())
    return [(b - a) * 1e3 for a, b in zip(t, t[1:])]


for _ in range(3):
    stage_times(False)
for sync in (True, False):
    acc = None
    for _ in range(5):
        r = stage_times(sync)
        acc = r if acc is None else [a + b for a, b in zip(acc, r)]
    print(shape, ("synchronised stages" if sync else "issue only        "), "encoder fwd %.2f | (proj +) joiner fwd %.2f | head step incl. ALL backward %.2f | drain %.2f  (ms)" % tuple(a / 5 for a in acc), flush=True)

if os.environ.get("W2VS_CPROFILE") == "1":
    import cProfile
    import pstats
    torch.autograd.set_multithreading_enabled(False)      # the backward nodes run on THIS thread: cProfile sees them
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        stage_times(False)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
