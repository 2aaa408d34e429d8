#!/usr/bin/env python
"""rocprofv3 (rocpd sqlite) result -> the durations (us) of every dispatch of the kernels whose name contains PATTERN, in start order.
usage: rocpd_calls.py <results.db> PATTERN [PATTERN ...]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
for pat in sys.argv[2:]:
    rows = c.execute("select (end - start), grid_size_x from kernels where name like ? order by start", ("%" + pat + "%",)).fetchall() \
        if "grid_size_x" in [r[1] for r in c.execute("pragma table_info(kernels)")] else \
        [(r[0], 0) for r in c.execute("select (end - start) from kernels where name like ? order by start", ("%" + pat + "%",)).fetchall()]
    print(pat, "dispatches in start order, us (grid x):", " ".join("%.0f(%d)" % (d / 1e3, g) for d, g in rows))
