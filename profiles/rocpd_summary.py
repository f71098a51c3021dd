#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default --kernel-trace --stats output of ROCm 7.2) as text."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
print("# rocprofv3 --kernel-trace --stats  (durations in microseconds)")
print("%-60s %6s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-60s %6d %14.3f %12.3f %7.3f" % (name.split("(")[0][:60], calls, total, avg, pct))
