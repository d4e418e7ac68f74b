#!/bin/bash
# Per-kernel totals of the LARGE configuration's step (bench.py --workload large) under rocprofv3 --kernel-trace --stats
mkdir -p gpurun_out; rm -rf gpurun_out/ktl
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/ktl -o prof --output-format csv -- python3 bench.py --workload large --steps 8 --warmup 4 --no-cpu-baseline --no-variants --no-gemm-peak > gpurun_out/ktl.log 2> gpurun_out/ktl.err || { echo failed; tail -5 gpurun_out/ktl.err; exit 1; }
f=$(find gpurun_out/ktl -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/large_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 12
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step %.2f ms" % (tot / steps / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-86s %6.1f/step avg %8.1f us  %7.1f us/step %5.1f%%" % (r["Name"][:86], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                                 float(r["TotalDurationNs"]) / steps / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf gpurun_out/ktl
