#!/usr/bin/env python3
"""BASELINE.json configs[3] on one GPU: wav2vec-S large (24 L, d = 1024, pre-LN, conv bias, 7 extractor LayerNorms),
3 x 320 000 samples (max_tokens 1.2 M), synthetic audio, random-init weights: ms per update with the fused Adam.
Not the headline metric; kept to show the large configuration runs on the same kernels."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import wav2vec_s_amd as w
    from wav2vec_s_amd import trainer
    import random
    import numpy as np
    cfg = w.large_librivox_config()
    torch.manual_seed(1)
    np.random.seed(1234)
    random.seed(1234)
    model = w.Wav2VecSModel(cfg).to(torch.bfloat16).cuda().train()
    crit = w.Wav2vecCriterion(infonce=True, loss_weights=[0.1, 0.0], log_keys=["prob_perplexity", "code_perplexity", "temp"])
    step = trainer.TrainStep(model, crit, lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, arena_gib=40.0)
    B, L = 3, 320000
    src = torch.randn(B, L, generator=torch.Generator().manual_seed(1234)).to(torch.bfloat16).cuda()
    src = torch.nn.functional.layer_norm(src.float(), (L,)).to(torch.bfloat16)       # task normalize: true
    sample = {"net_input": {"source": src}}
    for i in range(3):
        model.set_num_updates(i)
        step(sample)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for i in range(n):
        model.set_num_updates(3 + i)
        loss = step(sample)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    nparam = sum(p.numel() for p in model.parameters())
    print(json.dumps({"config": "wav2vec-S large, 3 x 320000 samples", "params_M": round(nparam / 1e6, 1), "ms_per_step": round(ms, 2),
                      "audio_s_per_s": round(B * L / 16000 / (ms / 1e3), 1), "loss": float(loss)}))


if __name__ == "__main__":
    main()
