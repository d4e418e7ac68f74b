#!/usr/bin/env python3
"""BASELINE configs[4] on one MI355X: streaming-ST fine-tune step = wav2vec-S base encoder twin (row f1) -> CAAT joiner
(6 layers, row f4) over decoder states -> TransducerOut (projection, delay transducer loss, cross entropy) -> backward through
all of it, at the shape the reference's training script runs (default) or at the `w2v2_caat` architecture defaults
(tools/caat_shapes.py describes both).

    python tools/bench_caat.py [--shape script|arch|both]
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
    ap.add_argument("--shape", default="script", choices=["script", "arch", "both"])
    ap.add_argument("--B", type=int, default=0)
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--U", type=int, default=0)
    ap.add_argument("--V", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "caat_bench.json"))
    a = ap.parse_args(argv)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import random
    import caat_shapes
    reps = {}
    for shape in (["script", "arch"] if a.shape == "both" else [a.shape]):
        random.seed(1234)                           # --step-mode random draws the group size from python's `random`
        built = caat_shapes.build(shape, a.B, a.samples, a.U, a.V)
        step, sh = built["step"], built["shape"]
        for _ in range(3):
            info, shp = step()
        torch.cuda.synchronize()
        n = 12
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        seen = {}
        ev[0].record()
        t0 = time.perf_counter()
        for i in range(n):
            info, shp = step()
            ev[i + 1].record()
            seen[built["jn"].downsample] = seen.get(built["jn"].downsample, 0) + 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        reps[shape] = {
            "workload": "BASELINE configs[4] (%s shape): wav2vec-S base encoder twin%s + 6-layer CAAT joiner %d / %d heads / ffn %d "
                        "(downsample %d, step-mode %s, dropout %.1f / %.1f / %.1f) + TransducerOut (%s delay, tokens_per_step %d), "
                        "fwd + bwd; decoder states synthetic" % (
                            shape, " + encoder_proj" if sh["D"] != 768 else "", sh["D"], sh["H"], sh["ffn"], sh["ds"],
                            sh["step_mode"], sh["dropout"], sh["act_dropout"], sh["attn_dropout"], sh["delay_func"],
                            sh["tokens_per_step"]),
            "shape": {"B": sh["B"], "samples": sh["samples"], "U": sh["U"], "V": sh["V"], "joint_last": list(shp),
                      "downsample_draws": {str(k): v for k, v in sorted(seen.items())}},
            "ms_per_step": round(ms, 3), "ms_per_step_gpu_median": round(sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2], 3),
            "audio_s_per_s": round(sh["B"] * sh["samples"] / 16000 / (ms / 1e3), 1), "loss": float(info["loss"])}
        del built, step
        torch.cuda.empty_cache()
    rep = reps[a.shape] if a.shape != "both" else reps
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    return rep


if __name__ == "__main__":
    main()
