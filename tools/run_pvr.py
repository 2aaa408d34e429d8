"""P4-scale PVR run: patch extraction (32x32 stride 16; `spx` = SLICO superpixel patches, --spxSize 32 --spxExtend 2 as in
BASELINE.json configs[4]), one outer iteration with 3 SR iterations; kernel times.  usage: run_pvr.py [spx|sq] [recon mm] [table]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from fetalreconstruction_amd import phantom, engine
from tests.twins import pvr

t0 = time.time()
RES = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
stacks, mask, mattr, rattr, rmask = phantom.make_stacks(4, (100, 93, 70), 1.17647, 1.25, 2.5, RES, 50.0, seed=1,
                                                        orientations=("ax", "cor", "sag", "ax"))
SPX = len(sys.argv) > 1 and sys.argv[1] == 'spx'
P = pvr.make_pvr_problem(stacks, mask, mattr, rattr, rmask, (32, 32), (2, 2) if SPX else (16, 16), superpixel=SPX)
print("patch extraction s:", round(time.time() - t0, 1), "patches", P.slices.shape, P.patches_per_stack,
      "non-zero px", int((P.slices > 0).sum()))
rec = engine.Reconstruction(0)
rec.set_option("pvr", 1)
if "table" in sys.argv:
    rec.set_option("coeff_table", 1)                     # the taps kept in HBM (12 units of 768 bytes per patch pixel)
engine.sync_gpu(rec, P, quality_factor=1.0)
if SPX:
    rec.set_spx_masks(P.spx_masks)
d = pvr.irtkPatchBasedReconstruction(rec, P.patches_per_stack, P.min_intensity, P.max_intensity)
rec.timer_enable(True)
t0 = time.time()
d.reconstruct_iteration(3)
print("outer iteration wall s:", round(time.time() - t0, 2))
for k, (ms, n) in rec.timers().items():
    if n:
        print(f"  {k}: {ms / n:.2f} ms x {n}")
c = rec.counters()
print("Va", c["Va"], "-> MVox/s per SR iteration (fwd+back):",
      round(c["Va"] / ((rec.timers()['backproject'][0] / rec.timers()['backproject'][1] + rec.timers()['forward'][0] / rec.timers()['forward'][1]) * 1e-3) / 1e6, 1))
