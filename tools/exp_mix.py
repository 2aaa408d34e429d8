"""Dev experiment (round 6): the MIXED pass -- `coeff_planes` of every 32 planes stream the coefficient table, the others' items
evaluate, in the same launch of the cell kernels -- against the two pure forms.  Same addon | cmap | simulated slices bit for bit?
Time per launch of the scatter and of the gather.  usage: exp_mix.py [P4|S8|PVR4|PVR8spx] [planes ...]   (planes: 0 = on the fly)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fetalreconstruction_amd import engine as E, workloads
from fetalreconstruction_amd.host import irtkReconstruction

name = sys.argv[1] if len(sys.argv) > 1 else "P4"
planes = [int(a) for a in sys.argv[2:]] or [0, 32, 24, 20, 16, 12]
reps = int(os.environ.get("REPS", "6"))
P = workloads.get(name)
pvr = name.startswith("PVR")
rec = E.Reconstruction(0)
if pvr:
    rec.set_option("pvr", 1)
    E.sync_gpu(rec, P, quality_factor=1.0)
    if getattr(P, "spx_masks", None) is not None:
        rec.set_spx_masks(P.spx_masks)
    ones = np.ones(P.ns, np.float32)
    rec.UpdateScaleVector(ones, ones)
    rec.InitializeEMValues()
    rec.GaussianReconstruction()
    rec.SimulateSlices()
else:
    E.sync_gpu(rec, P)
    drv = irtkReconstruction(rec, P.ns, (0, P.ns), None, P.max_intensity, P.min_intensity)
    drv.SetSmoothingParameters(150, 0.02)
    drv.InitializeEMValuesGPU(); drv.GaussianReconstructionGPU(); drv.SimulateSlicesGPU(); drv.InitializeRobustStatisticsGPU(); drv.EStepGPU()
    for i in range(int(os.environ.get("ITERS", "3"))):
        drv.sr_iteration(i)
ones = np.ones(P.ns, np.float32)
ref = None
rec.timer_enable(True)
for m in planes:
    rec.set_option("coeff_table", 1 if m else 0)
    if m:
        rec.set_option("coeff_planes", m)
    if os.environ.get("FWD_MODE"):
        rec.set_option("fwd_mode", int(os.environ["FWD_MODE"]))
    rec.SimulateSlices()
    rec.SuperresolutionBackproject(ones)
    rec.timer_reset()
    for _ in range(reps):
        rec.SuperresolutionBackproject(ones)
        rec.SimulateSlices()
    t = rec.timers()
    out = (rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy(), rec.debug_get(E.BUF_SIMSLICES).copy(), rec.debug_get(E.BUF_SIMWEIGHTS).copy())
    same = "-" if ref is None else str([bool(np.array_equal(a, b, equal_nan=True)) for a, b in zip(out, ref)])
    if ref is None:
        ref = out
    bt = t["backproject"][0] / t["backproject"][1]
    ft = t["forward"][0] / t["forward"][1]
    cb = t.get("coeff_build", (0.0, 0))
    print(f"[{name}] planes {m:2d}/32 (table on: {rec.get_option('coeff_table')}): scatter {bt:.3f} ms, gather {ft:.3f} ms, sum {bt + ft:.3f}; "
          f"build {cb[0] / max(cb[1], 1):.3f} ms x{cb[1]}; same bits as the first line {same}", flush=True)
rec.close()
