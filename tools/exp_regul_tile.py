import json, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from fetalreconstruction_amd import workloads, engine, host
for wl in sys.argv[1:]:
    P = workloads.get(wl)
    pvr = wl.startswith("PVR")
    rec = engine.Reconstruction(0)
    if pvr:
        rec.set_option("pvr", 1); engine.sync_gpu(rec, P, quality_factor=1.0)
        if getattr(P, "spx_masks", None) is not None: rec.set_spx_masks(P.spx_masks)
        d = host.irtkPatchBasedReconstruction(rec, P.patches_per_stack, P.min_intensity, P.max_intensity); d.reconstruct_iteration(1); sw = d.state()["patch_weight"]
        args = (False, 0.5, P.min_intensity, P.max_intensity, 1.0, 0.1)
    else:
        engine.sync_gpu(rec, P)
        d = host.irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity); d.SetSmoothingParameters(150, 0.02); d.reconstruct_iteration(1); sw = d.state()["slice_weight"]
        args = (False, 0.8, P.min_intensity, P.max_intensity, 150.0, 0.02 * 150.0 ** 2)
    v0 = rec.syncCPU().copy()
    rec.SuperresolutionBackproject(sw)
    rec.timer_enable(True)
    out = {"workload": wl}
    for tile in (-1, 0, 1, 2):
        rec.set_option("reg_tile", tile)
        for rep in range(12):
            if rep == 2: rec.timer_reset()
            rec.SuperresolutionUpdate(*args)
        t = rec.timers()
        out["tile%d" % tile] = round(t["regularize"][0] / t["regularize"][1], 4)
    print(json.dumps(out), flush=True)
