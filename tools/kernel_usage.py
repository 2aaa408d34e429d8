#!/usr/bin/env python3
"""Compact table of hipcc's -Rpass-analysis=kernel-resource-usage remarks for the engine's kernels.

usage: python tools/kernel_usage.py [-DNAME=VALUE ...] [--filter substr]
Compiles csrc/svr_hip.hip to a throw-away object (nothing in lib/ is touched)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fetalreconstruction_amd import build as B  # noqa: E402


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
    with tempfile.TemporaryDirectory() as td:
        cmd = [B.hipcc(), *[f for f in B.FLAGS if f != "-shared"], "-c", "-Rpass-analysis=kernel-resource-usage", *defs,
               "-o", os.path.join(td, "x.o"), B.SRC]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode:
            sys.stderr.write(p.stderr[-4000:])
            sys.exit(p.returncode)
    rows, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            dem = re.sub(r"\(anonymous namespace\)::", "", dem)
            dem = re.sub(r"\(.*$", "", dem)
            cur = {"name": dem}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
    for r in rows:
        if flt and flt not in r["name"]:
            continue
        print(f"{r['name'][:58]:58s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('SGPRs', '?'):>5s} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('Occupancy [waves/SIMD]', '?'):>4s} "
              f"{r.get('LDS Size [bytes/block]', '?'):>7s}")


if __name__ == "__main__":
    main()
