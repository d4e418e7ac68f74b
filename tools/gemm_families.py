#!/usr/bin/env python3
"""Per-kernel-family GEMM throughput INSIDE the training step (libw2vs's own event samples, ops.GEMM_TIMER ids): which
family is far from its isolated-probe rate.  python tools/gemm_families.py [stride]"""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for k, v in (("OMP_NUM_THREADS", "4"), ("OMP_WAIT_POLICY", "PASSIVE"), ("GOMP_SPINCOUNT", "0"), ("MKL_NUM_THREADS", "4")):
    os.environ.setdefault(k, v)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import wav2vec_s_amd as w  # noqa: E402
from wav2vec_s_amd import _lib, trainer  # noqa: E402

stride = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda", 0)
cfg = w.base_librispeech_config()
torch.manual_seed(1)
model = w.Wav2VecSModel(cfg).to(torch.bfloat16).to(dev).train()
crit = w.Wav2vecCriterion(infonce=True, loss_weights=[0.1, 10.0])
step = trainer.TrainStep(model, crit)
sample = {"net_input": {"source": torch.randn(8, 175000).to(torch.bfloat16).to(dev)}}
np.random.seed(1234); random.seed(1234); torch.manual_seed(1234)
for i in range(4):
    step(sample)
torch.cuda.synchronize()
_lib.call("w2vs_prof_enable", stride)
for i in range(12):
    step(sample)
torch.cuda.synchronize()
epis = ["none", "bias", "bias_gelu", "bias_gelu_save", "dgelu", "f32", "add", "bias_gelu_savegrad", "mul"]
names = {16 * f + e: "%s<%s>" % (fn, en) for f, fn in enumerate(("gemm_nt_kernel", "gemm_nt_lc_kernel", "gemm_nt_p_kernel"))
         for e, en in enumerate(epis)}
names.update({10: "gemm_tn_kernel", 11: "gemm_tn_lc_kernel(+reduce)", 12: "gemm_tn_group_kernel"})
rows = []
for k in sorted(names):
    ms, fl, n = C.c_double(), C.c_double(), C.c_int32()
    _lib.call("w2vs_prof_read", k, C.byref(ms), C.byref(fl), C.byref(n))
    if n.value:
        launches = int(_lib.load().w2vs_prof_launches(k))
        est_ms_step = ms.value / n.value * launches / 12
        rows.append((est_ms_step, names[k], n.value, ms.value / n.value * 1e3, fl.value / n.value / 1e9, fl.value / ms.value / 1e9, launches / 12))
rows.sort(reverse=True)
print("%-36s %7s %9s %9s %8s %8s %9s" % ("family", "samples", "avg us", "GF/launch", "TF/s", "per step", "ms/step"))
for est, name, n, us, gf, tf, per in rows:
    print("%-36s %7d %9.1f %9.1f %8.0f %8.1f %9.3f" % (name, n, us, gf, tf, per, est))
_lib.call("w2vs_prof_enable", 0)
