#!/usr/bin/env python3
"""Row f3 measurement (input side).

    python tools/bench_data.py [--out gpurun_out/data_bench.json]

* batcher: `w2vs_batch_by_size` (C ABI, host) on a LibriSpeech-960-sized epoch (281 241 utterances, max_tokens
  1.4 M, multiple of 8); the reference's own compiled Cython (oracle/_ref) is timed beside it only through
  `python bench.py --workload data` (bench.py's cpu_baseline leg owns every use of oracle/);
* collater: `w2vs_collate` at the pre-training batch shape (8 utterances of 11-16 s -> [8, 175 000] bf16, with and
  without whole-utterance normalisation): kernel time from HIP events on the launching stream with inputs resident
  (HBM-bound: algorithmic bytes = 4 B x all samples for the statistics + 4 B read and 2 B written per output sample),
  and the PCIe-inclusive time of the whole collater call.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None, cpu_baseline=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "data_bench.json"))
    a = ap.parse_args(argv)
    from wav2vec_s_amd import _lib, data
    rep = {}
    rng = np.random.RandomState(0)
    n = 281241
    sizes = np.minimum((np.clip(rng.gamma(6.0, 2.3, size=n), 1.3, 35.0) * 16000).astype(np.int64), 250000)
    sizes = np.sort(sizes)[::-1].copy()
    idx = np.arange(n, dtype=np.int64)
    t0 = time.perf_counter()
    for _ in range(5):
        b = data.batch_by_size_vec(idx, sizes, 1400000, -1, 8)
    t = (time.perf_counter() - t0) / 5
    rep["batch_by_size"] = {"n": n, "batches": len(b), "ms": round(t * 1e3, 3), "utterances_per_s": round(n / t)}
    if cpu_baseline is not None:                    # bench.py's cpu_baseline leg: the reference's compiled Cython batcher
        rep["batch_by_size"].update(cpu_baseline(idx, sizes, b))

    g = torch.Generator().manual_seed(3)
    lens = [250000, 176000, 243111, 175000, 201234, 199999, 180001, 250000]
    waves = [torch.randn(m, generator=g) * 0.05 for m in lens]
    target = min(lens)
    starts = [0 if m <= target else (m - target) // 2 for m in lens]
    for norm in (False, True):
        # PCIe-inclusive: the whole call (pinned staging, H2D, kernels)
        for _ in range(3):
            data.collate_device(waves, target, starts, pad=False, normalize=norm, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            data.collate_device(waves, target, starts, pad=False, normalize=norm, device="cuda")
        torch.cuda.synchronize()
        t_call = (time.perf_counter() - t0) / 20
        # kernels only, inputs resident
        dev = torch.device("cuda")
        B = len(lens)
        sz = np.array(lens, dtype=np.int32)
        offs = np.zeros(B, dtype=np.int64); offs[1:] = np.cumsum(sz[:-1].astype(np.int64))
        flat = torch.cat(waves).to(dev)
        off_d, sz_d = torch.tensor(offs, device=dev), torch.tensor(sz, device=dev)
        st_d = torch.tensor(starts, dtype=torch.int32, device=dev)
        out = torch.empty(B, target, dtype=torch.bfloat16, device=dev)
        lib = _lib.load()
        nch = int(lib.w2vs_collate_chunks(int(sz.max())))
        part = torch.empty(B * nch * 2, dtype=torch.float64, device=dev)
        d = _lib.CollateDesc()
        d.flat, d.offset, d.size, d.crop_start = flat.data_ptr(), off_d.data_ptr(), sz_d.data_ptr(), st_d.data_ptr()
        d.out, d.padding_mask, d.partial = out.data_ptr(), None, part.data_ptr()
        d.B, d.target, d.width, d.max_size, d.normalize, d.out_f32 = B, target, target, int(sz.max()), int(norm), 0
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            _lib.call("w2vs_collate", C.byref(d), stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            _lib.call("w2vs_collate", C.byref(d), stream)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        alg = (int(sz.sum()) * 4 if norm else 0) + B * target * (4 + 2)
        rep["collate_norm" if norm else "collate"] = {
            "batch": [B, target], "kernel_us": round(us, 2), "algorithmic_MB": round(alg / 1e6, 2),
            "achieved_GBps": round(alg / us / 1e3, 1), "hbm_peak_GBps": 8000,
            "call_ms_pcie_inclusive": round(t_call * 1e3, 3),
            "audio_s_per_s_pcie_inclusive": round(B * target / 16000 / t_call)}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
