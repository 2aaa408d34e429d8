"""Dev check: after a few SR iterations of the bench's driver, the scatter with and without the coefficient table -- same
addon | cmap?  how many pixels carry non-zero factors?  time per launch.  usage: check_table_scatter.py [P4|PVR4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fetalreconstruction_amd import engine as E, workloads
from fetalreconstruction_amd.host import irtkReconstruction

name = sys.argv[1] if len(sys.argv) > 1 else "P4"
P = workloads.get(name)
rec = E.Reconstruction(0)
E.sync_gpu(rec, P)
drv = irtkReconstruction(rec, P.ns, (0, P.ns), None, P.max_intensity, P.min_intensity)
drv.SetSmoothingParameters(150, 0.02)
drv.InitializeEMValuesGPU(); drv.GaussianReconstructionGPU(); drv.SimulateSlicesGPU(); drv.InitializeRobustStatisticsGPU(); drv.EStepGPU()
for i in range(int(os.environ.get("ITERS", "25"))):
    drv.sr_iteration(i)
w, sw = rec.debug_get(E.BUF_WEIGHTS), rec.debug_get(E.BUF_SIMSLICES)
act = P.slices != -1
print(f"[{name}] pixels s != -1: {int(act.sum())}, weight == 0: {int((w[act] == 0).sum())}, simslices <= 0: {int((sw[act] <= 0).sum())}")
ones = np.ones(P.ns, np.float32)
res = {}
for tab in (0, 1):
    rec.set_option("coeff_table", tab)
    rec.SimulateSlices()
    rec.SuperresolutionBackproject(ones)
    rec.timer_enable(True); rec.timer_reset()
    for _ in range(5):
        rec.SuperresolutionBackproject(ones)
    t = rec.timers()["backproject"]
    res[tab] = (rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy())
    print(f"[{name}] table {rec.get_option('coeff_table')} back_mode {rec.get_option('back_mode')}: scatter {t[0] / t[1]:.3f} ms", flush=True)
rel = lambda a, b: float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
print(f"[{name}] table vs on the fly: addon {rel(res[1][0], res[0][0]):.1e} cmap {rel(res[1][1], res[0][1]):.1e}, hit sets equal {np.array_equal(res[1][1] > 0, res[0][1] > 0)}")
