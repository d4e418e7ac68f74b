#!/usr/bin/env python3
"""Per (kernel symbol, workgroup count) medians from a rocprofv3 kernel-trace CSV: the per-family averages of a --stats file mix
shapes (conv and encoder launches of one GEMM symbol), this separates them.
    python tools/trace_shapes.py gpurun_out/final_prof/prof_kernel_trace.csv [steps] [filter]"""
import collections
import csv
import re
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
flt = sys.argv[3] if len(sys.argv) > 3 else ""
agg = collections.defaultdict(list)
for r in rows:
    n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("w2vs::", "").replace("(anonymous namespace)::", "").split("(")[0]
    if flt and flt not in n:
        continue
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
    agg[(n, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = sorted(((sum(v) / steps, n, g, len(v) / steps, statistics.median(v)) for (n, g), v in agg.items()), reverse=True)
print("%-46s %6s %9s %11s %12s" % ("kernel", "wgs", "per step", "median us", "us per step"))
for t, n, g, c, m in out[:70]:
    print("%-46s %6d %9.1f %11.1f %12.1f" % (n[:46], g, c, m, t))
print("total us per step: %.1f" % sum(o[0] for o in out))
