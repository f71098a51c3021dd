#!/usr/bin/env python3
"""Steady-state kernel timeline from a rocprofv3 rocpd database: every kernel that starts between the frontend
dispatch of step n-4 and that of step n-2 (n = last), with its queue / stream id where the database has one.
usage: rocpd_steps.py <db> [first_back=4] [last_back=2]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
fb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
extra = [x for x in ("queue_id", "stream_id", "queue", "stream") if x in cols]
q = "select name,start,end" + "".join("," + x for x in extra) + " from kernels order by start"
rows = list(c.execute(q))
fr = [i for i, r in enumerate(rows) if "frontend_kernel" in r[0]]
a, b = rows[fr[-fb]][1], rows[fr[-lb]][1]
print("# columns of kernels table:", cols)
print("%-34s %9s %9s %8s  %s" % ("kernel", "start_ms", "end_ms", "dur_ms", " ".join(extra)))
for r in rows:
    if a <= r[1] < b:
        print("%-34s %9.3f %9.3f %8.3f  %s" % (r[0].split("(")[0].replace("tfrec::", "").replace("void ", "")[:34], (r[1] - a) / 1e6,
                                               (r[2] - a) / 1e6, (r[2] - r[1]) / 1e6, " ".join(str(x) for x in r[3:])))
print("# interval: %.3f ms for %d steps" % ((b - a) / 1e6, fb - lb))
