"""Dev experiment: the scatter without atomics (back_mode 5) against mode 4 -- parity on tiny, timing and run-to-run identity on
the bench workloads.  usage: exp_cell.py [P4|S8|PVR4 ...] [table]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fetalreconstruction_amd import engine as E, phantom, workloads

def timed_scatter(rec, ns, reps=5):
    ones = np.ones(ns, np.float32)
    rec.SuperresolutionBackproject(ones)           # warm-up / tuning / cell_prepare
    rec.timer_enable(True); rec.timer_reset()
    for _ in range(reps):
        rec.SuperresolutionBackproject(ones)
    ms, n = rec.timers()["backproject"]
    return ms / n

def setup(P, pvr=False, table=False):
    rec = E.Reconstruction(0)
    if pvr:
        rec.set_option("pvr", 1)
        E.sync_gpu(rec, P, quality_factor=1.0)
        if getattr(P, "spx_masks", None) is not None:
            rec.set_spx_masks(P.spx_masks)
    else:
        E.sync_gpu(rec, P)
    if table:
        rec.set_option("coeff_table", 1)
    ones = np.ones(P.ns, np.float32)
    rec.UpdateScaleVector(ones, ones)
    rec.InitializeEMValues()
    return rec

OPTS = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a)}
sys.argv = [a for a in sys.argv if "=" not in a]
names = [a for a in sys.argv[1:] if a not in ("table", "gather")] or ["P4"]
table = "table" in sys.argv
for name in names:
    P = workloads.get(name) if name != "tiny" else phantom.problem_tiny()
    pvr = name.startswith("PVR")
    res = {}
    for mode in (4, 5):
        rec = setup(P, pvr, table)
        rec.set_option("back_mode", mode)
        for k, v in OPTS.items():
            rec.set_option(k, v)
        t0 = time.time(); rec.GaussianReconstruction(); tg = time.time() - t0
        t0 = time.time(); rec.GaussianReconstruction(); tg2 = time.time() - t0
        vol, vw = rec.syncCPU().copy(), rec.getVolWeights().copy()
        rng = np.random.default_rng(0)
        rec.debug_set(E.BUF_SIMSLICES, np.where(P.slices > 0, P.slices * rng.uniform(0.8, 1.2, P.slices.shape), 0).astype(np.float32))
        rec.debug_set(E.BUF_WEIGHTS, np.where(P.slices != -1, rng.uniform(0.2, 1.0, P.slices.shape), 0).astype(np.float32))
        ms = timed_scatter(rec, P.ns)
        if "gather" in sys.argv:
            for fm in (1, 2):
                rec.set_option("fwd_mode", fm)
                rec.SimulateSlices(); rec.timer_reset()
                for _ in range(5):
                    rec.SimulateSlices()
                t = rec.timers()["forward"]
                print(f"[{name}] fwd_mode {fm}: gather {t[0] / t[1]:.3f} ms", flush=True)
        a1, c1 = rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy()
        rec.SuperresolutionBackproject(np.ones(P.ns, np.float32))
        a2, c2 = rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy()
        res[mode] = (vol, vw, a1, c1)
        print(f"[{name}{' table' if table else ''}] {OPTS} back_mode {mode}: scatter {ms:.3f} ms, gaussian pass wall {tg*1e3:.1f} / {tg2*1e3:.1f} ms, "
              f"run-to-run identical: {np.array_equal(a1, a2) and np.array_equal(c1, c2)}, table on: {rec.get_option('coeff_table')}", flush=True)
        rec.close()
    r4, r5 = res[4], res[5]
    rel = lambda a, b: float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
    print(f"[{name}] mode 5 vs 4: recon {rel(r5[0], r4[0]):.1e} volw {rel(r5[1], r4[1]):.1e} addon {rel(r5[2], r4[2]):.1e} cmap {rel(r5[3], r4[3]):.1e}; "
          f"hit sets equal: {np.array_equal(r5[3] > 0, r4[3] > 0)} {np.array_equal(r5[1] > 0, r4[1] > 0)}", flush=True)
