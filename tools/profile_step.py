#!/usr/bin/env python3
"""Host-side profile of the training step: issue time vs GPU time, cProfile of the Python path."""
import cProfile
import os
import pstats
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for k, v in (("OMP_NUM_THREADS", "4"), ("OMP_WAIT_POLICY", "PASSIVE"), ("GOMP_SPINCOUNT", "0"), ("MKL_NUM_THREADS", "4")):
    if os.environ.get("W2VS_NO_OMP_LIMIT") is None:
        os.environ.setdefault(k, v)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import wav2vec_s_amd as w  # noqa: E402
from wav2vec_s_amd import trainer  # noqa: E402

dev = torch.device("cuda", 0)
cfg = w.base_librispeech_config()
torch.manual_seed(1)
model = w.Wav2VecSModel(cfg).to(torch.bfloat16).to(dev).train()
crit = w.Wav2vecCriterion(infonce=True, loss_weights=[0.1, 10.0])
step = trainer.TrainStep(model, crit)
src = torch.randn(8, 175000).to(torch.bfloat16).to(dev)
sample = {"net_input": {"source": src}}
np.random.seed(1234); random.seed(1234); torch.manual_seed(1234)
for i in range(3):
    step(sample)
torch.cuda.synchronize()
for i in range(8):
    t0 = time.perf_counter()
    step(sample)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = model._last_state
    print("step %d: issue %.2f ms, total %.2f ms  (N=%d m=%d r=%d kept=%d)" % (i, (t1 - t0) * 1e3, (t2 - t0) * 1e3, st.N, st.m, st.r, len(st.kept)), flush=True)
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    step(sample)
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr).sort_stats("cumulative")
ps.print_stats(35)
