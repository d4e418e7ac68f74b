#!/bin/bash
# Average duration of the kernels matching $1 (regex) in a short bench.py run under rocprofv3 --kernel-trace --stats, for
# the library in $2 (default: the product).  Same invocation form as tools/collect_profiles.sh.
PAT=${1:?kernel name regex}; LIB=${2:-wav2vec-s_amd/libw2vs.so}
mkdir -p gpurun_out; rm -rf gpurun_out/kt
W2VS_LIB=$PWD/$LIB timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o prof --output-format csv -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-variants --no-gemm-peak > gpurun_out/kt.log 2> gpurun_out/kt.err || { echo failed; tail -5 gpurun_out/kt.err; exit 1; }
f=$(find gpurun_out/kt -name "*kernel_stats.csv" | head -1)
echo "$LIB:"; python3 - "$f" "$PAT" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("  %-90s calls %5s  avg %9.1f us  total %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
rm -rf gpurun_out/kt
