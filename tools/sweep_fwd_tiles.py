"""forward-gather tile sweep: time SimulateSlices for fwd tile sizes / LDS box capacities on a workload.
usage: sweep_fwd_tiles.py [P4|S8|S8h]"""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from fetalreconstruction_amd import phantom, engine
from fetalreconstruction_amd.reconstruction import irtkReconstruction
w = sys.argv[1] if len(sys.argv) > 1 else "S8h"
if w == "P4":
    P = phantom.problem_p4()
elif w == "S8":
    P = phantom.problem_s8()
else:
    P = phantom.make_problem(8, (256, 256, 64), 1.0, 2.5, 2.5, 0.5, 100.0, orientations=("ax", "cor", "sag"), name="S8h")
rec = engine.Reconstruction(0); engine.sync_gpu(rec, P)
d = irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity); d.SetSmoothingParameters(150, 0.02)
d.InitializeEMValuesGPU(); d.GaussianReconstructionGPU(); d.SimulateSlicesGPU()
ref = rec.debug_get(engine.BUF_SIMSLICES).copy()
rec.timer_enable(True)
for tw, th, cap in [(8, 4, 9216), (8, 4, 13600), (8, 4, 18000), (4, 4, 9216), (4, 4, 13600), (4, 2, 9216), (8, 2, 9216), (8, 2, 13600), (2, 2, 9216), (8, 8, 18000)]:
    rec.set_option("fwd_tile_w", tw); rec.set_option("fwd_tile_h", th); rec.set_option("fwd_cap", cap)
    rec.SimulateSlices()
    rec.timer_reset()
    for _ in range(2):
        rec.SimulateSlices()
    ms, n = rec.timers()["forward"]
    err = float(np.abs(rec.debug_get(engine.BUF_SIMSLICES) - ref).max())
    print(f"{w} tile {tw}x{th} cap {cap}: forward {ms / n:.2f} ms  (max |diff| vs default {err:.3g})", flush=True)
