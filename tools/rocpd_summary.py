#!/usr/bin/env python
"""Turns a rocprofv3 (ROCm 7.x rocpd sqlite) result into the text kernel-stats summary that is
committed under profiles/.  usage: rocpd_summary.py <results.db> [title]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    print(f"# rocprofv3 --kernel-trace --stats : {title}")
    print(f"{'calls':>7} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel")
    for name, calls, tot, avg, pct in rows:
        print(f"{calls:7d} {tot:14.1f} {avg:12.1f} {pct:7.2f}  {name}")
    try:
        cur = c.execute("select name, vgpr_count, sgpr_count, lds_block_size, workgroup_size_x, grid_size_x "
                        "from kernels group by name")
        print("\n# per-kernel launch resources (first dispatch)")
        for r in cur:
            print("  ", r)
    except Exception:
        pass


if __name__ == "__main__":
    main()
