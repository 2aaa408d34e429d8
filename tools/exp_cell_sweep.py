"""Dev experiment: scatter / gather time of the cell kernels against the cell size (all pixels weighted).
usage: exp_cell_sweep.py WORKLOAD "WxH,WxH,..." [opt=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fetalreconstruction_amd import engine as E, workloads

name, sizes = sys.argv[1], sys.argv[2]
opts = {k: int(v) for k, v in (a.split("=") for a in sys.argv[3:])}
P = workloads.get(name)
pvr = name.startswith("PVR")
rec = E.Reconstruction(0)
if pvr:
    rec.set_option("pvr", 1)
    E.sync_gpu(rec, P, quality_factor=1.0)
    if getattr(P, "spx_masks", None) is not None:
        rec.set_spx_masks(P.spx_masks)
else:
    E.sync_gpu(rec, P)
for k, v in opts.items():
    rec.set_option(k, v)
ones = np.ones(P.ns, np.float32)
rec.UpdateScaleVector(ones, ones)
rec.InitializeEMValues()
rec.GaussianReconstruction()
rec.SimulateSlices()
rng = np.random.default_rng(0)
rec.debug_set(E.BUF_SIMSLICES, np.where(P.slices > 0, P.slices * rng.uniform(0.8, 1.2, P.slices.shape), 0).astype(np.float32))
rec.debug_set(E.BUF_WEIGHTS, np.where(P.slices != -1, rng.uniform(0.2, 1.0, P.slices.shape), 0).astype(np.float32))
rec.timer_enable(True)
for s in sizes.split(","):
    w, h = (int(x) for x in s.split("x"))
    for k, v in (("cell_w", w), ("cell_h", h), ("cell_gw", w), ("cell_gh", h)):
        rec.set_option(k, v)
    rec.SuperresolutionBackproject(ones); rec.SimulateSlices()
    rec.timer_reset()
    for _ in range(5):
        rec.SuperresolutionBackproject(ones)
        rec.SimulateSlices()
    t = rec.timers()
    print(f"[{name}] cell {w}x{h}: scatter {t['backproject'][0] / t['backproject'][1]:.3f} ms, gather {t['forward'][0] / t['forward'][1]:.3f} ms", flush=True)
