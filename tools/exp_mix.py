"""Dev experiment (round 6): the forms of the two PSF passes on one workload -- coeff_table 0 (every tap evaluated) and 1 (the taps of a live unit streamed
from the coefficient table; the gather through the LDS) -- timed per launch and compared bit for bit (addon | cmap | simulated slices / weights), plus the
gather that evaluates and writes the table against k_coeff_build.  usage: exp_mix.py [P4|S8|PVR4|PVR8spx] [modes ...] [option=value ...]
(the wavefront-mix, half-table and shared-box variants this script first measured are recorded in profiles/r06_plane_mix_experiment.txt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fetalreconstruction_amd import engine as E, workloads
from fetalreconstruction_amd.host import irtkReconstruction

OPTS = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a)}
sys.argv = [a for a in sys.argv if "=" not in a]
name = sys.argv[1] if len(sys.argv) > 1 else "P4"
planes = [int(a) for a in sys.argv[2:]] or [0, 1]
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
    rec.set_option("coeff_table", m)
    for k, v in OPTS.items():
        rec.set_option(k, v)
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
    if m:                                   # the gather that writes the table (coeff_lazy), and the separate build (k_coeff_build)
        rec.timer_reset(); rec.set_option("coeff_invalidate", 1); rec.SimulateSlices()
        st_ms = rec.timers()["forward"][0]
        rec.set_option("coeff_lazy", 0); rec.timer_reset(); rec.set_option("coeff_invalidate", 1); rec.SimulateSlices()
        tt = rec.timers(); bl_ms = tt["coeff_build"][0]; rec.set_option("coeff_lazy", 1)
        print(f"[{name}] the gather that writes the table: {st_ms:.3f} ms; k_coeff_build: {bl_ms:.3f} ms", flush=True)
    bt = t["backproject"][0] / t["backproject"][1]
    ft = t["forward"][0] / t["forward"][1]
    cb = t.get("coeff_build", (0.0, 0))
    print(f"[{name}] {OPTS} cells {rec.get_option('cell_w')}x{rec.get_option('cell_h')} / {rec.get_option('cell_gw')}x{rec.get_option('cell_gh')} coeff_table {m} (in effect: {rec.get_option('coeff_table')}): scatter {bt:.3f} ms, gather {ft:.3f} ms, sum {bt + ft:.3f}; "
          f"build {cb[0] / max(cb[1], 1):.3f} ms x{cb[1]}; same bits as the first line {same}", flush=True)
rec.close()
