#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd database:  python tools/kstats.py <results.db> [min_calls]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
tot, cnt = collections.Counter(), collections.Counter()
for n, s, e in rows:
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "").split("(")[0][:80]
    tot[n] += e - s
    cnt[n] += 1
for n, t in tot.most_common(60):
    print("%-82s %6d calls %10.1f us total %9.2f us avg" % (n, cnt[n], t / 1e3, t / cnt[n] / 1e3))
