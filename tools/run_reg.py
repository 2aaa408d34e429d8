"""P4-scale run of the GPU slice-to-volume registration: reconstruct, prepare, register; prints time and
counters, with the batched gradient and one evaluation at a time.  usage: run_reg.py [tiny|p4]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from fetalreconstruction_amd import phantom, engine
from tests.twins import registration as R
from tests.twins.reconstruction import irtkReconstruction

which = sys.argv[1] if len(sys.argv) > 1 else "p4"
P = phantom.problem_p4() if which == "p4" else phantom.problem_tiny()
rec = engine.Reconstruction(0)
engine.sync_gpu(rec, P)
d = irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
d.SetSmoothingParameters(150, 0.02)
d.reconstruct_iteration(2)
t0 = time.time()
rs = R.PrepareRegistrationSlices(rec, P.slices, P.slice_attr, P.vdim[0])
print("prep (numpy resampling) s:", round(time.time() - t0, 2), "grid", rs.combined.shape)
T = P.slice_t.reshape(-1, 4, 4).astype(np.float64)
rec.timer_enable(True)
res = {}
for batch, blind in ((1, 4), (1, 8), (1, 0), (0, 0)):
    rec.set_option("reg_batch", batch)
    rec.set_option("reg_blind", blind)
    for rep in range(2):
        rec.timer_reset()
        t0 = time.time()
        Tn = R.SliceToVolumeRegistrationGPU(rec, rs, T)
        wall = time.time() - t0
        c = rec.reg_counters()
        ms, n = rec.timers()["register"]
        print(f"reg_batch {batch} reg_blind {blind} rep {rep}: wall {wall:.3f} s, device-timer {ms:.1f} ms, evaluations {c[0]}, line-search {c[1]}, "
              f"iterations {c[2]}, slice-evaluations {c[3]} -> {c[3] / wall / 1e3:.1f} k slice-evals/s, "
              f"{c[3] * 3 * rs.combined.shape[1] * rs.combined.shape[2] / wall / 1e9:.2f} G samples/s", flush=True)
    res[(batch, blind)] = (Tn, c)
print("max |dT|", np.abs(res[(1, 4)][0] - T).max(), " default vs literal: max |dT|", np.abs(res[(1, 4)][0] - res[(0, 0)][0]).max(),
      "counters equal:", bool(np.array_equal(res[(1, 4)][1], res[(0, 0)][1])))
