"""per-pixel kernel times over reconstruction resolutions (voxels per pixel from 0.67 to 2.9): looks for cliffs where a tile's
volume box stops fitting LDS.  4 stacks of 128x128x32, 1.0 mm pixels."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from fetalreconstruction_amd import phantom, engine
from tests.twins.reconstruction import irtkReconstruction
for res in (1.5, 1.0, 0.75, 0.5, 0.4, 0.35):
    P = phantom.make_problem(4, (128, 128, 32), 1.0, 2.5, 2.5, res, 50.0, orientations=("ax", "cor", "sag"), name="r")
    rec = engine.Reconstruction(0); engine.sync_gpu(rec, P)
    d = irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity); d.SetSmoothingParameters(150, 0.02)
    d.InitializeEMValuesGPU(); d.GaussianReconstructionGPU(); d.SimulateSlicesGPU(); d.InitializeRobustStatisticsGPU(); d.EStepGPU()
    rec.timer_enable(True); rec.timer_reset()
    d.GaussianReconstructionGPU()
    for i in range(2):
        d.sr_iteration(i)
    t = rec.timers(); c = rec.counters()
    va = c["Va"]
    print(f"recon {res} mm: volume {P.vsize}, Va {va}: back {t['backproject'][0] / t['backproject'][1] / va * 1e6:.2f} ns/px, "
          f"forward {t['forward'][0] / t['forward'][1] / va * 1e6:.2f} ns/px, gauss {t['gauss'][0] / max(t['gauss'][1], 1) / va * 1e6:.2f} ns/px", flush=True)
    rec.close()
