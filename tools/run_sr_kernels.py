"""dev tool: a few launches of the scatter and the gather on a workload with the given options (for rocprofv3 passes).
usage: python tools/run_sr_kernels.py [workload] [opt=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fetalreconstruction_amd import workloads, engine, host
args = sys.argv[1:]
wl = args.pop(0) if args and '=' not in args[0] else 'P4'
P = workloads.get(wl)
pvr = wl.startswith("PVR")
rec = engine.Reconstruction(0)
if pvr:
    rec.set_option("pvr", 1)
    engine.sync_gpu(rec, P, quality_factor=1.0)
    if getattr(P, "spx_masks", None) is not None:
        rec.set_spx_masks(P.spx_masks)
else:
    engine.sync_gpu(rec, P)
for a in args:
    k, v = a.split('='); rec.set_option(k, int(v))
if pvr:
    d = host.irtkPatchBasedReconstruction(rec, P.patches_per_stack, P.min_intensity, P.max_intensity)
    d.reconstruct_iteration(1)
    sw = d.state()["patch_weight"]
else:
    d = host.irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity); d.SetSmoothingParameters(150, 0.02)
    d.reconstruct_iteration(1)                     # the state of a running reconstruction: EM weights, a non-trivial residual
    sw = d.state()["slice_weight"]
for _ in range(40):          # many more than the tuners' trial launches: pmc.py averages the last half of a kernel's dispatches
    rec.SuperresolutionBackproject(sw)
upd = (False, 0.5, float(P.min_intensity), float(P.max_intensity), 1.0, 0.1) if pvr else (False, 0.8, float(P.min_intensity), float(P.max_intensity), 150.0, 0.02 * 150.0 ** 2)
v0 = rec.syncCPU().copy()
for _ in range(20):          # the volume update (k_regul_fused) on the scatter's addon | cmap
    rec.SuperresolutionUpdate(*upd)
    rec.SuperresolutionBackproject(sw)
rec.UpdateReconstructed(rec.vsize, v0)
for _ in range(40):
    rec.SimulateSlices()
if not pvr and rec.get_option("coeff_table") == 1:
    for _ in range(8):       # the gather that evaluates and WRITES the coefficient table (coeff_lazy): once per new slice geometry in a run, eight times here
        rec.set_option("coeff_invalidate", 1)
        rec.SimulateSlices()
    for _ in range(8):       # ... and the scatter that does (when it is the first PSF pass after the table went)
        rec.set_option("coeff_invalidate", 1)
        rec.SuperresolutionBackproject(sw)
print("tuned: scatter mode %d tiles %dx%d box %d, gather tiles %dx%d box %d" % tuple(rec.get_option(k) for k in ("back_mode", "tile_w", "tile_h", "wave_cap", "fwd_tile_w", "fwd_tile_h", "fwd_unit_cap")), flush=True)
