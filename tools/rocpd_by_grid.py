#!/usr/bin/env python
"""rocprofv3 (rocpd sqlite) result -> per kernel AND grid size: calls, average duration.  Separates the launches of the whole
workload from the launches of one rank's range when both run in one process (tools/shard_probe.py).
usage: rocpd_by_grid.py <results.db> [title]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    dur = "(end - start)"
    g = [x for x in ("grid_size_x", "grid_size_y", "grid_size_z", "grid_x", "grid_y", "grid_z") if x in cols][:3]
    if len(g) < 3:
        print("columns of `kernels`:", cols)
        g = ["0", "0", "0"]
    rows = list(c.execute(f"select name, {g[0]}, {g[1]}, {g[2]}, count(*), avg({dur}), sum({dur}) from kernels "
                          f"group by name, {g[0]}, {g[1]}, {g[2]} order by sum({dur}) desc"))
    print(f"# rocprofv3 --kernel-trace : {sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]} -- per kernel and grid")
    print(f"{'calls':>6} {'avg_us':>11} {'total_us':>12}  {'grid':>22}  kernel")
    for name, gx, gy, gz, n, avg, tot in rows:
        if tot / 1e3 < 20:
            continue
        print(f"{n:6d} {avg / 1e3:11.1f} {tot / 1e3:12.1f}  {str((gx, gy, gz)):>22}  {short(name)}")


if __name__ == "__main__":
    main()
