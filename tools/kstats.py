#!/usr/bin/env python
"""Print average kernel durations (us) from a rocprofv3 results .db, optionally filtered by substring."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for name, calls, avg in c.execute("select name,total_calls,average from top_kernels"):
    if pat in name:
        print(f"{name.replace('(anonymous namespace)::', '')[:60]:60s} x{calls:5d}  {avg:8.1f} us")
