"""dev tool (for rocprofv3 --kernel-trace --stats): the scatter and the gather of ONE rank's range of a workload, nothing else.
usage: python tools/shard_one.py WORKLOAD RANK WORLD [reps] [opt=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fetalreconstruction_amd import phantom
from fetalreconstruction_amd.sharding import patch_cost_weights, shard_slices, slice_cost_weights
from tools.shard_probe import build, make_engine

wl, r, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rest = sys.argv[4:]
reps = int(rest.pop(0)) if rest and "=" not in rest[0] else 10
opts = [(o.split("=")[0], int(o.split("=")[1])) for o in rest]
prob = build(wl)
pvr = wl.startswith("PVR")
if pvr:
    work = patch_cost_weights((prob.slices > 0).reshape(prob.ns, -1).sum(1), prob.slice_i2w, prob.slice_t, prob.recon_w2i)
else:
    act = (prob.slices != -1).reshape(prob.ns, -1).sum(1)
    work = slice_cost_weights(act, prob.slice_i2w, prob.slice_t, prob.recon_w2i, prob.slice_dim, prob.vdim[0])
lo, hi = shard_slices(work, world)[r]
sub = phantom.sub_problem(prob, lo, hi)
spx = getattr(prob, "spx_masks", None)
rec = make_engine(sub, pvr, None if spx is None else spx[lo:hi], opts)
ones = np.ones(sub.ns, np.float32)
rec.UpdateScaleVector(ones, ones)
rec.InitializeEMValues()
rec.GaussianReconstruction()
rec.SimulateSlices()
rec.timer_enable(True)
for k in range(reps + 1):
    if k == 1:
        rec.timer_reset()
    rec.SuperresolutionBackproject(ones)
    rec.SimulateSlices()
t = rec.timers()
print(wl, "rank", r, "of", world, "units", (lo, hi), "Va", rec.counters()["Va"], {k: round(v[0] / max(v[1], 1), 3) for k, v in t.items() if v[1]}, rec.cell_stats())
