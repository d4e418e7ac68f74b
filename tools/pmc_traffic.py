#!/usr/bin/env python3
"""HBM-side bytes per launch from two rocprofv3 counter passes (rocpd databases):
    python tools/pmc_traffic.py <fetch.db> <write.db> > profiles/<name>.json
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B by rocprofv3 (counter x 64 B / 1024); on gfx950
FETCH_SIZE counts a 128-B request as 64 B, so reads are doubled (MI355X_MICROARCH.md, HBM section)."""
import collections
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    tot, cnt = collections.Counter(), collections.Counter()
    q = "select %s, counter_name, sum(value) from counters_collection where counter_name = ? group by dispatch_id" % name_col
    for name, _, v in db.execute(q, (counter,)):
        name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").split("(")[0]
        tot[name] += v
        cnt[name] += 1
    return tot, cnt


fetch, fc = per_kernel(sys.argv[1], "FETCH_SIZE")
write, wc = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch, key=lambda k: -fetch[k]):
    out[k] = {"launches": fc[k],
              "read_MB_per_launch_corrected": round(2.0 * fetch[k] * 1024 / fc[k] / 1e6, 2),
              "write_MB_per_launch": round(write.get(k, 0.0) * 1024 / max(wc.get(k, 1), 1) / 1e6, 2)}
# bench.py quotes this file only while csrc/gemm.hip is the build it was collected from
import hashlib
import os
_src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wav2vec-s_amd", "csrc", "gemm.hip")
out["_gemm_hip_sha256"] = hashlib.sha256(open(_src, "rb").read()).hexdigest()[:16]
json.dump(out, sys.stdout, indent=1)
