"""Dev check: how the bench's state moves over its SR iterations (slice weights, pixel weights), and what a scatter costs in it."""
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
act = P.slices != -1
it = 0
rec.timer_enable(True)
for upto in (2, 12, 24, 34, 46, 60):
    rec.timer_reset()
    while it < upto:
        drv.sr_iteration(it); it += 1
    st = drv.state()
    sw = st["slice_weight"]
    w = rec.debug_get(E.BUF_WEIGHTS)
    t = rec.timers()
    print(f"[{name}] after {it:3d} iterations: slice_weight == 0: {int((sw == 0).sum())} < 0.01: {int((sw < 0.01).sum())} of {P.ns}, mean {sw.mean():.3f}; pixel weight == 0: {int((w[act] == 0).sum())}; "
          f"scatter {t['backproject'][0] / max(t['backproject'][1], 1):.3f} ms gather {t['forward'][0] / max(t['forward'][1], 1):.3f} ms", flush=True)
