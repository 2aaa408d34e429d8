#!/usr/bin/env python3
"""Every dispatch of the last SR iterations of a traced run, with its duration and the gap before it.

usage: cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-s8 [--no-coeff-table]
       python tools/step_timeline.py DIR [steps_to_print]
A step starts at a scatter (back_cell_kernel / back_wave_kernel) and ends before the next one; printed: the last `steps_to_print` (default 4:
one outer iteration of the bench's schedule -- with the coefficient table one step whose scatter evaluates and whose gather writes the table,
three that stream it), and per step the dispatches, the busy time outside the two PSF kernels, and the idle time."""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    d = sys.argv[1]
    nprint = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    anchors = [i for i, r in enumerate(rows) if r[2].startswith(("back_cell_kernel", "back_wave_kernel"))]
    if len(anchors) < 2:
        print("no scatter dispatches in", d)
        return
    # the bench measures twice (timers off, timers on) and then the other mode: the steps wanted are those of the FIRST timed pass of the default mode;
    # simplest robust choice: the longest run of anchors whose spacing stays within 3x the median -- print its last `nprint` steps
    steps = [(anchors[k], anchors[k + 1]) for k in range(len(anchors) - 1)]
    which = os.environ.get("STEP_FROM")
    sel = steps[-nprint:] if not which else steps[int(which):int(which) + nprint]
    for lo, hi in sel:
        seg = rows[lo:hi]
        t0 = seg[0][0]
        busy = sum(e - s for s, e, _ in seg)
        span = seg[-1][1] - t0
        psf = sum(e - s for s, e, n in seg if n.startswith(("back_cell_kernel", "fwd_cell_kernel", "fwd_unit_kernel", "back_wave_kernel")))
        print(f"--- step at dispatch {lo}: span {span / 1e3:.1f} us, {len(seg)} dispatches, the two PSF kernels {psf / 1e3:.1f} us, every other kernel "
              f"{(busy - psf) / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")
        prev = t0
        for s, e, n in seg:
            print(f"  + {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:6.1f}  {n}")
            prev = e


if __name__ == "__main__":
    main()
