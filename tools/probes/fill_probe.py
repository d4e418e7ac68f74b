#!/usr/bin/env python3
"""L2 -> LDS fill rate per CU against the bytes in flight (tools/probes/fill_probe.hip).  python tools/probes/fill_probe.py"""
import ctypes as C
import os
import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "fill_probe.so"))
lib.fill_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
region = 4 << 20
src = torch.randint(0, 255, (8 * region,), dtype=torch.uint8, device="cuda")
cyc = torch.zeros(1024, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
chunks = 4000


def run(grid, depth, bar):
    for _ in range(2):
        rc = lib.fill_probe(src.data_ptr(), region, chunks, depth, bar, cyc.data_ptr(), grid, st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        lib.fill_probe(src.data_ptr(), region, chunks, depth, bar, cyc.data_ptr(), grid, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    return chunks * 16384 / (ms * 1e-3) / 1e9          # GB/s per workgroup (= per CU when grid <= 256)


for grid in (256, 64, 8):
    for bar in (0, 1):
        row = []
        for d in (1, 2, 3, 4, 6, 8, 9):
            row.append("%d:%5.1f" % (d * 16, run(grid, d, bar)))
        print("grid %3d  barrier %d   KiB in flight : GB/s per CU   %s" % (grid, bar, "  ".join(row)), flush=True)


lib.fill_probe_waves.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
for bar in (0, 1):
    row = []
    for w in (1, 2, 4, 8):
        for _ in range(2):
            assert lib.fill_probe_waves(src.data_ptr(), region, chunks, w, bar, cyc.data_ptr(), 256, st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.fill_probe_waves(src.data_ptr(), region, chunks, w, bar, cyc.data_ptr(), 256, st)
        e1.record()
        torch.cuda.synchronize()
        row.append("%d waves:%6.1f" % (w, chunks * 16384 / (e0.elapsed_time(e1) * 1e-3) / 1e9))
    print("grid 256, 48 KiB in flight, barrier %d, issuing waves per CU : GB/s per CU   %s" % (bar, "  ".join(row)), flush=True)
