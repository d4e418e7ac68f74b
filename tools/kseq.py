#!/usr/bin/env python3
"""Durations (us) of every launch whose kernel name contains <filter>, in launch order: python tools/kseq.py <db> <filter> [group]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = [(e - s) / 1e3 for n, s, e in db.execute("select name, start, end from kernels order by start") if sys.argv[2] in n]
grp = int(sys.argv[3]) if len(sys.argv) > 3 else 10
for i in range(0, len(rows), grp):
    c = rows[i:i + grp]
    print("%4d: med %.1f  " % (i, sorted(c)[len(c) // 2]) + " ".join("%.1f" % v for v in c))
