#!/usr/bin/env python3
"""The kernels of ONE streaming call (B = 1, prefix seconds from argv, eval, eager launches) - run under
rocprofv3 --kernel-trace --stats to see where the ~1.6 ms of a call go.   python tools/stream_kernels.py [seconds] [calls]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wav2vec_s_amd import streaming  # noqa: E402
from wav2vec_s_amd.config import base_librispeech_config  # noqa: E402

sec = int(sys.argv[1]) if len(sys.argv) > 1 else 10
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = base_librispeech_config(main_context=16, right_context=8, context_type="constant")
torch.manual_seed(1)
model = streaming.BlockWiseWav2Vec2Model(cfg).to(torch.bfloat16).cuda().eval()
model.graph_calls = False
s1 = torch.randn(1, sec * 16000).to(torch.bfloat16).cuda()
with torch.no_grad():
    for _ in range(calls):
        model(s1, None, None, False, True)
torch.cuda.synchronize()
print("done", calls, "calls")
