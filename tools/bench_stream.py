#!/usr/bin/env python3
"""Row f1 measurement: the streaming / fine-tune encoder twin on one MI355X.

    python tools/bench_stream.py [--steps 20] [--warmup 5] [--out gpurun_out/stream_bench.json]

Reports (synthetic audio, random-init base model, bf16, eval mode, m=16 / r=8):
  * offline batch throughput at the pre-training batch shape (8 x 175 000 samples), forward only and
    forward+backward (fine-tuning with the extractor's feature_grad_mult=0.1);
  * per-call latency of the SimulEval-style call pattern, where every new main-context block re-encodes the whole
    prefix (rain/layers/unidirect_w2v2_encoder.py:245-330 has no incremental state): B=1 at 2 / 10 / 30 s prefixes.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stream_bench.json"))
    a = ap.parse_args(argv)
    from wav2vec_s_amd import streaming
    from wav2vec_s_amd.config import base_librispeech_config
    cfg = base_librispeech_config(main_context=16, right_context=8, context_type="constant")
    torch.manual_seed(1)
    model = streaming.BlockWiseWav2Vec2Model(cfg).to(torch.bfloat16).cuda()
    g = torch.Generator().manual_seed(1234)
    rep = {"model": "wav2vec-S base twin (12 x 768), bf16, m=16 r=8", "data": "synthetic", "steps": a.steps}

    B, L = 8, 175000
    src = torch.randn(B, L, generator=g).to(torch.bfloat16).cuda()
    model.eval()
    model.graph_calls = True                   # the replay path is opt-in: measured here beside the eager one
    with torch.no_grad():
        ms = timed(lambda: model(src), a.steps, a.warmup)
        model.graph_calls = False
        ms_e = timed(lambda: model(src), a.steps, a.warmup)
        model.graph_calls = True
    rep["offline_fwd"] = {"batch": [B, L], "ms": round(ms, 3), "ms_eager": round(ms_e, 3),
                          "audio_s_per_s": round(B * L / 16000 / (ms / 1e3), 1)}

    model.train()
    w = None

    def step():
        nonlocal w
        x = model(src)["encoder_out"][0]
        if w is None:
            w = torch.randn_like(x)
        model.zero_grad(set_to_none=True)
        (x * w).sum().backward()

    ms = timed(step, max(a.steps // 2, 3), 3)
    rep["finetune_fwd_bwd"] = {"batch": [B, L], "ms": round(ms, 3),
                               "audio_s_per_s": round(B * L / 16000 / (ms / 1e3), 1),
                               "note": "dropout 0.1 / attention_dropout 0.1 / LayerDrop 0.05 active; gradients handed "
                                       "to autograd per parameter (no flat arena)"}

    model.eval()
    rep["streaming_call"] = []
    for sec in (2, 10, 30):
        s1 = torch.randn(1, sec * 16000, generator=g).to(torch.bfloat16).cuda()
        with torch.no_grad():
            ms = timed(lambda: model(s1, None, None, False, True), a.steps, a.warmup)      # one HIP graph per shape, replayed
            model.graph_calls = False
            ms_e = timed(lambda: model(s1, None, None, False, True), a.steps, a.warmup)    # the same call launched kernel by kernel
            model.graph_calls = True
        rep["streaming_call"].append({"prefix_s": sec, "ms_per_call": round(ms, 3), "ms_per_call_eager": round(ms_e, 3),
                                      "block_ms": 16 * 20, "real_time_factor": round(ms / (16 * 20), 4)})
    # a growing-prefix session (what a streaming client does): 320 ms chunks up to 10 s, four utterances of the same chunk
    # schedule.  With graph_after = 2 the first two run eagerly, the third captures (the miss cost), the fourth replays.
    chunk = 5120
    n_chunks = 31
    fresh = streaming.BlockWiseWav2Vec2Model(cfg).to(torch.bfloat16).cuda().eval()
    fresh.load_state_dict(model.state_dict())
    fresh.graph_calls = True
    per_utt = []
    with torch.no_grad():
        for ui in range(4):
            u = torch.randn(1, chunk * n_chunks, generator=g).to(torch.bfloat16).cuda()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for c in range(2, n_chunks + 1):
                fresh(u[:, :c * chunk].contiguous(), None, None, False, True)
            torch.cuda.synchronize()
            per_utt.append(round((time.perf_counter() - t0) * 1e3 / (n_chunks - 1), 3))
    gs = fresh.graph_stats()
    rep["growing_prefix_session"] = {"chunk_ms": 320, "utterance_s": 9.92, "calls_per_utterance": n_chunks - 1,
                                     "ms_per_call_by_utterance": per_utt, "graph_after": fresh.graph_after,
                                     "hits": gs["hits"], "misses": gs["misses"], "captures": gs["captures"],
                                     "graphs_held": gs["graphs"], "bytes_held": gs["bytes"],
                                     "note": "utterances 1-2 eager, 3 = capture (miss cost), 4 = replay"}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
