#!/usr/bin/env python3
"""LDS read rate per CU by instruction (tools/probes/fill_probe.hip: lds_read_kernel).  python tools/probes/lds_probe.py"""
import ctypes as C
import os
import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "fill_probe.so"))
lib.lds_read_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
iters = 20000
names = {0: "ds_read_b128 linear", 3: "ds_read_b128 swizzled", 1: "ds_read_b64", 2: "ds_read_b64_tr_b16"}
bytes_per = {0: 16, 3: 16, 1: 8, 2: 8}
for mode in (0, 3, 1, 2):
    for waves in (4, 8):
        for _ in range(2):
            assert lib.lds_read_probe(mode, waves, iters, sink.data_ptr(), 256, st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.lds_read_probe(mode, waves, iters, sink.data_ptr(), 256, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        byts = waves * iters * 8 * 64 * bytes_per[mode]
        print("%-24s %d waves/CU: %6.1f GB/s per CU  (= %.0f B/clk at 2.4 GHz)" % (names[mode], waves, byts / ms / 1e6, byts / ms / 1e6 / 2.4), flush=True)
