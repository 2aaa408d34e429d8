"""P4-scale run of the default (IRTK schedule) slice-to-volume registration with every similarity on the GPU: reconstruct,
mask, register the 280 slices in lock step; prints wall time and evaluation counts.  usage: run_irtk_reg.py [tiny|p4]"""
import sys
import time

sys.path.insert(0, '/root/repo')
import numpy as np  # noqa: E402

from fetalreconstruction_amd import engine, geometry as geo, host, phantom  # noqa: E402
from tests.twins.reconstruction import irtkReconstruction  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "p4"
P = phantom.problem_p4() if which == "p4" else phantom.problem_tiny()
rec = engine.Reconstruction(0)
engine.sync_gpu(rec, P)
d = irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
d.SetSmoothingParameters(150, 0.02)
d.reconstruct_iteration(4)
vol = rec.syncCPU().reshape(P.vsize[::-1])
rattr = geo.ImageAttributes(*P.vsize, *P.vdim)
T = P.slice_t.reshape(-1, 4, 4).astype(np.float64)
for rep in range(2):
    t0 = time.time()
    Tn, nev = host.SliceToVolumeRegistration(rec, P.slices, P.slice_attr, T, rattr, vol)
    wall = time.time() - t0
    print(f"rep {rep}: {P.ns} slices, wall {wall:.3f} s, {nev} similarity evaluations -> {nev / wall / 1e3:.1f} k evaluations/s", flush=True)
p = np.concatenate([np.random.default_rng(0).uniform(-30, 30, (200, 3)), np.ones((200, 1))], 1)
move = [float(np.linalg.norm((p @ a.T - p @ b.T)[:, :3], axis=1).max()) for a, b in zip(Tn, T)]
print("slice displacement mm: median", round(float(np.median(move)), 2), "max", round(float(np.max(move)), 2))
