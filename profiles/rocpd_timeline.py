#!/usr/bin/env python3
"""Print the kernel timeline (start offset / duration, ms) of the LAST step in a rocprofv3 rocpd database.
A step starts at the last tfrec::frontend_kernel dispatch."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = list(c.execute("select name,start,end,grid_x,grid_y,workgroup_x from kernels order by start")) if "grid_x" in cols else \
    [r + (0, 0, 0) for r in c.execute("select name,start,end from kernels order by start")]
last = max(i for i, r in enumerate(rows) if "frontend_kernel" in r[0])
t0 = rows[last][1]
print("%-40s %10s %10s %10s  grid" % ("kernel", "start_ms", "end_ms", "dur_ms"))
for name, s, e, gx, gy, wx in rows[last:]:
    print("%-40s %10.3f %10.3f %10.3f  %sx%s/%s" % (name.split("(")[0][:40], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, gx, gy, wx))
