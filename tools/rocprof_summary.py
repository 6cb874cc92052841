#!/usr/bin/env python
"""Turn a rocprofv3 `--kernel-trace --stats` result database (rocpd sqlite) into the per-kernel summary table that is
committed under profiles/.  Usage: python tools/rocprof_summary.py <results.db> <out.md> [title]"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out, "w") as f:
        f.write(f"# {title}\n\nSource: rocprofv3 --kernel-trace --stats (durations in microseconds).\n\n")
        f.write("| kernel | calls | total us | avg us | % of GPU time |\n|---|---:|---:|---:|---:|\n")
        for name, calls, tot, avg, pct in rows[:60]:
            f.write(f"| `{str(name)[:110]}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |\n")
    print(f"wrote {out} ({len(rows)} kernels)")


if __name__ == "__main__":
    main()
