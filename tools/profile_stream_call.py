#!/usr/bin/env python3
"""Where does a streaming call (B = 1, 10 s prefix, eval) spend its time?  Host issue time vs total, cProfile of the host path,
and the number of kernel launches (torch profiler).   python tools/profile_stream_call.py"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from wav2vec_s_amd import streaming  # noqa: E402
from wav2vec_s_amd.config import base_librispeech_config  # noqa: E402

cfg = base_librispeech_config(main_context=16, right_context=8, context_type="constant")
torch.manual_seed(1)
model = streaming.BlockWiseWav2Vec2Model(cfg).to(torch.bfloat16).cuda().eval()
s1 = torch.randn(1, 160000).to(torch.bfloat16).cuda()
with torch.no_grad():
    for _ in range(5):
        model(s1, None, None, False, True)
    torch.cuda.synchronize()
    for i in range(5):
        t0 = time.perf_counter()
        model(s1, None, None, False, True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("call %d: issue %.3f ms, total %.3f ms" % (i, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        model(s1, None, None, False, True)
    e1.record()
    torch.cuda.synchronize()
    print("20 calls back to back: %.3f ms per call (GPU timeline)" % (e0.elapsed_time(e1) / 20))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        model(s1, None, None, False, True)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
