#!/usr/bin/env python3
"""Cuts a rocprofv3 run of bench.py by the training step's roctx ranges (trainer.ROCTX: forward / backward / reduce-grads /
clip-grads / optimizer - the names of fairseq's own profiler ranges, fs/trainer.py:754-795, fs/tasks/fairseq_task.py:474-478).

    rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace -d DIR -o p --output-format csv -- python3 bench.py ...
    python tools/phase_cut.py DIR/p

A range brackets the ENQUEUE of its kernels on the host; the kernels themselves run later.  Every launch API record inside a
range carries the correlation id of the kernel it started, so the GPU time of a phase is the summed duration of those kernels."""
import csv
import sys
from collections import defaultdict

base = sys.argv[1]


def rows(name):
    with open(base + name) as f:
        return list(csv.DictReader(f))


markers = rows("_marker_api_trace.csv")
api = rows("_hip_api_trace.csv")
kern = rows("_kernel_trace.csv")
kdur = defaultdict(float)
for k in kern:
    kdur[k["Correlation_Id"]] += (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3
launches = sorted((int(a["Start_Timestamp"]), a["Correlation_Id"]) for a in api if "Launch" in a["Function"])
ranges = [(m["Function"], int(m["Start_Timestamp"]), int(m["End_Timestamp"])) for m in markers if int(m["End_Timestamp"]) > int(m["Start_Timestamp"])]
tot, cnt, nk = defaultdict(float), defaultdict(int), defaultdict(int)
import bisect
starts = [t for t, _ in launches]
for name, t0, t1 in ranges:
    i, j = bisect.bisect_left(starts, t0), bisect.bisect_right(starts, t1)
    tot[name] += sum(kdur[c] for _, c in launches[i:j])
    nk[name] += j - i
    cnt[name] += 1
print("%-14s %8s %14s %16s" % ("range", "count", "kernels/range", "GPU us / range"))
for name in sorted(tot, key=lambda n: -tot[n]):
    print("%-14s %8d %14.1f %16.1f" % (name, cnt[name], nk[name] / cnt[name], tot[name] / cnt[name]))
