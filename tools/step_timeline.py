"""Dev tool: the kernel timeline of the last SR iterations of a bench run, from a rocprofv3 --kernel-trace database:
busy time, gaps between kernels, per-kernel share of a step.  usage (on the GPU box):
  cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d DIR -o t -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-coeff-table
  python tools/step_timeline.py DIR [anchor-kernel-substring]"""
import glob, os, re, sqlite3, sys

d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "back_cell_kernel"
db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:60]
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
if len(idx) < 4:
    raise SystemExit("fewer than 4 anchor kernels")
# a step = from one anchor launch to the next; the last three complete steps
for s in range(len(idx) - 4, len(idx) - 1):
    a, b = idx[s], idx[s + 1]
    t0 = rows[a][1]
    span = rows[b][1] - t0
    busy = sum(r[2] - r[1] for r in rows[a:b])
    print(f"--- step {s}: span {span / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us, {b - a} dispatches")
    prev_end = t0
    for r in rows[a:b]:
        gap = r[1] - prev_end
        print(f"  +{(r[1] - t0) / 1e3:9.1f} us  dur {(r[2] - r[1]) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  {short(r[0])}")
        prev_end = r[2]
    print(f"  gap to the next step's anchor: {(rows[b][1] - prev_end) / 1e3:.1f} us")
