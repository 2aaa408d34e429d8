"""dev tool: the volume update (Prep + regulariser) of a workload timed in both forms, and compared.
usage: python tools/run_regul.py [workload ...]   -> one JSON line per workload"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fetalreconstruction_amd import workloads, engine, host

for wl in (sys.argv[1:] or ["P4"]):
    P = workloads.get(wl)
    pvr = wl.startswith("PVR")
    rec = engine.Reconstruction(0)
    if pvr:
        rec.set_option("pvr", 1)
        engine.sync_gpu(rec, P, quality_factor=1.0)
        if getattr(P, "spx_masks", None) is not None:
            rec.set_spx_masks(P.spx_masks)
        d = host.irtkPatchBasedReconstruction(rec, P.patches_per_stack, P.min_intensity, P.max_intensity)
        d.reconstruct_iteration(1)
        sw = d.state()["patch_weight"]
        args = (False, 0.5, P.min_intensity, P.max_intensity, 1.0, 0.1)
    else:
        engine.sync_gpu(rec, P)
        d = host.irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity); d.SetSmoothingParameters(150, 0.02)
        d.reconstruct_iteration(1)
        sw = d.state()["slice_weight"]
        args = (False, 0.8, P.min_intensity, P.max_intensity, 150.0, 0.02 * 150.0 ** 2)   # reconstruction.cc:118-121 defaults
    v0 = rec.syncCPU().copy()
    rec.SuperresolutionBackproject(sw)
    out = {"workload": wl, "Nv": int(v0.size), "dims": [int(v) for v in rec.vsize]}
    vols = {}
    for mode in (0, 1):
        rec.set_option("reg_mode", mode)
        rec.timer_enable(True)
        for rep in range(12):
            rec.UpdateReconstructed(rec.vsize, v0)
            if rep == 2:
                rec.timer_reset()
            rec.SuperresolutionUpdate(*args)
        t = rec.timers()
        vols[mode] = rec.syncCPU().copy()
        ms = t["regularize"][0] / max(1, t["regularize"][1])
        out["mode%d_ms" % mode] = round(ms, 4)
        out["mode%d_frac_of_24B_per_voxel_at_8TBs" % mode] = round(24.0 * v0.size / (ms * 1e-3) / 8e12, 4)
    out["max_rel_diff"] = float(np.max(np.abs(vols[0] - vols[1])) / np.max(np.abs(vols[0])))
    out["nonzero_sets_equal"] = bool(np.array_equal(vols[0] != 0, vols[1] != 0))
    print(json.dumps(out), flush=True)
    rec.close()
