"""Dev experiment: what lies between two wavefronts in one hardware slot.  SVR_CELL_TRACE=file makes the scatter write, per entry,
{start, end (s_memrealtime, 100 MHz), HW_ID | XCC_ID << 32}; the last launch's trace is analysed here.
usage: wave_gaps.py WORKLOAD"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

wl = sys.argv[1]
trace = "/tmp/cell_trace.bin"
os.environ["SVR_CELL_TRACE"] = trace
from tools.shard_probe import build, make_engine
from fetalreconstruction_amd import engine as E
P = build(wl)
pvr = wl.startswith("PVR")
rec = make_engine(P, pvr, getattr(P, "spx_masks", None))
ones = np.ones(P.ns, np.float32)
rec.UpdateScaleVector(ones, ones); rec.InitializeEMValues(); rec.GaussianReconstruction(); rec.SimulateSlices()
rng = np.random.default_rng(0)
rec.debug_set(E.BUF_WEIGHTS, np.where(P.slices != -1, rng.uniform(0.2, 1.0, P.slices.shape), 0).astype(np.float32))
for _ in range(3):
    rec.SuperresolutionBackproject(ones)
t = np.fromfile(trace, np.uint64).reshape(-1, 3)
st, en, hw = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2]
t0 = st.min()
st -= t0; en -= t0
tick = 0.01                                               # us per tick
print(f"{wl}: {len(t)} entries, launch {en.max() * tick:.1f} us, mean wavefront life {np.mean(en - st) * tick:.1f} us, sum of lives {np.sum(en - st) * tick / 1e3:.2f} ms")
# slot = (xcc, se, sh, cu, simd, wave)
key = hw
order = np.lexsort((st, key))
k, s, e = key[order], st[order], en[order]
same = k[1:] == k[:-1]
gap = (s[1:] - e[:-1])[same]
slots = len(np.unique(key))
print(f"  hardware slots seen {slots} (HW_ID | XCC_ID distinct values); gaps between consecutive wavefronts of a slot: n {len(gap)}, "
      f"mean {gap.mean() * tick:.2f} us, median {np.median(gap) * tick:.2f}, p90 {np.percentile(gap, 90) * tick:.2f}, negative {int((gap < 0).sum())}")
# occupancy over time: resident wavefronts sampled
T = en.max()
grid = np.linspace(0, T, 200)
res = [(int(((st <= g) & (en > g)).sum())) for g in grid]
print("  resident wavefronts at 10 %, 50 %, 90 % of the launch:", res[20], res[100], res[180], " max", max(res))
first_start = np.sort(st)[:slots]
print(f"  the first {slots} wavefronts start within {first_start.max() * tick:.1f} us; the launch's last start at {st.max() * tick:.1f} us, last end {T * tick:.1f} us")
bits = np.bitwise_or.reduce(hw)
print("  HW_ID bits seen: %x" % int(bits))
