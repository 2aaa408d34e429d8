#!/usr/bin/env python3
"""rocprofv3 counter passes over a command, summarised per kernel.

usage: python tools/pmc.py OUTDIR [--filter substr] [--by-grid] [--sets "A B C" "D E" ...] -- <command ...>
(--by-grid: launches of one kernel with different grid sizes are reported apart, e.g. the whole workload and one rank's range)
Each set is one rocprofv3 pass (--kernel-trace --pmc <set>, nothing else: gpurun refuses --pmc next to the sys/hip/hsa
trace domains).  Per kernel (short name) the mean over the LAST HALF of its dispatches is printed: the first dispatches
hold warm-up and tile-shape timing.  Durations come from the same passes (End - Start of the dispatch)."""
import collections
import csv
import glob
import hashlib
import os
import re
import subprocess
import sys

DEFAULT_SETS = [
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT",
    "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD",
    "FETCH_SIZE",
    "WRITE_SIZE",
]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:60]


def main():
    argv = sys.argv[1:]
    out = os.path.abspath(argv.pop(0))
    cmd = argv[argv.index("--") + 1:]
    argv = argv[:argv.index("--")]
    flt = argv[argv.index("--filter") + 1] if "--filter" in argv else ""
    sets = DEFAULT_SETS
    by_grid = "--by-grid" in argv
    if "--sets" in argv:
        i = argv.index("--sets") + 1
        sets = []
        while i < len(argv) and not argv[i].startswith("--"):
            sets.append(argv[i])
            i += 1
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    for s in sets:
        tag = hashlib.md5(s.encode()).hexdigest()[:6]
        d = os.path.join(out, "pass_" + tag)
        r = subprocess.run(["timeout", "600", "rocprofv3", "--kernel-trace", "--pmc", *s.split(), "--output-format", "csv", "-d", d, "-o", "p",
                            "--", *cmd], cwd="/tmp", env=env, capture_output=True, text=True)
        open(os.path.join(out, f"pass_{tag}.log"), "w").write(r.stdout[-4000:] + r.stderr[-4000:])
        if r.returncode:
            print(f"# pass failed ({r.returncode}): {s}")
    res = collections.defaultdict(lambda: collections.defaultdict(dict))     # kernel -> counter -> dispatch -> value
    dur = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(out, "pass_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if flt and flt not in k:
                continue
            if by_grid:
                k += "  grid " + r.get("Grid_Size", "?")
            did = int(r["Dispatch_Id"])
            res[k][r["Counter_Name"]][did] = res[k][r["Counter_Name"]].get(did, 0.0) + float(r["Counter_Value"])
            dur[k][(f, did)] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k in sorted(res):
        ds = sorted(dur[k].values())
        n = len(next(iter(res[k].values())))
        print(f"{k}   dispatches/pass {n}   duration under profiling: median {ds[len(ds) // 2]:.1f} us")
        for c in sorted(res[k]):
            v = [res[k][c][d] for d in sorted(res[k][c])]
            v = v[len(v) // 2:]
            print(f"   {c:26s} {sum(v) / len(v):14.5g}   (mean of last {len(v)})")


if __name__ == "__main__":
    main()
