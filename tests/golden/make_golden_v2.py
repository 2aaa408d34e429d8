"""Regenerates tests/golden/tiny_v2_reg_pvr.npz from the CPU oracle: fixed targets for the rows
added after tiny_v1 -- GPU slice-to-volume registration (a17/f1) and the PVR loop (a18).  Like tiny_v1 these are the oracle's own outputs on seeded phantoms (the reference ships no
vectors); they pin the oracle against drift and give the GPU tests targets that do
not depend on the oracle being rebuilt.  Run from the repo root:  python tests/golden/make_golden_v2.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from fetalreconstruction_amd import geometry as geo  # noqa: E402
from fetalreconstruction_amd import phantom  # noqa: E402
from tests.twins import pvr  # noqa: E402
from tests.twins import registration as R  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def analytic_volume(P, radius):
    vx, vy, vz = P.vsize
    kk, jj, ii = np.meshgrid(np.arange(vz), np.arange(vy), np.arange(vx), indexing="ij")
    w = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(float) @ P.recon_i2w.reshape(4, 4).astype(float).T
    vol = phantom.phantom_intensity(w[..., :3], radius) * 700 / 0.55
    return np.where(P.mask > 0, vol, -1).astype(np.float32)


class _Rec:
    def initRegStorageVolumes(self, *a):
        pass

    def FillRegSlices(self, d, m):
        pass


def registration_case():
    P = phantom.problem_tiny()
    vol = analytic_volume(P, 14.0)
    rs = R.PrepareRegistrationSlices(_Rec(), P.slices, P.slice_attr, P.vdim[0])
    o = po.OracleRegistration(P.vsize, P.vdim[0], P.recon_w2i)
    o.initRegStorageVolumes(rs.combined.shape[2], rs.combined.shape[1], P.ns)
    o.FillRegSlices(rs.combined, rs.i2w)
    T = P.slice_t.reshape(-1, 4, 4).astype(np.float64)
    T[3] = T[3] @ geo.rigid_matrix(tx=1.5, rz=2.0)
    T[10] = T[10] @ geo.rigid_matrix(ty=-1.0, rx=-1.5)
    Tn = R.SliceToVolumeRegistrationGPU(o, rs, T, vol)
    mo = [np.eye(4) for _ in range(P.ns)]
    for m, a in zip(mo, rs.attrs):
        m[:3, 3] = a.origin
    t_in = np.stack([geo.to_matrix4(t @ m) for t, m in zip(T, mo)])
    sims = np.stack([o.evaluate_costs(t_in, lv)[0] for lv in (0, 1)])
    few = o.evaluate_costs(t_in, 0, active=[2, 5, 7])[0]
    return {"reg_combined_sum": np.float64(rs.combined.astype(np.float64).sum()), "reg_t_out": Tn.astype(np.float32),
            "reg_counters": o.counters.copy(), "reg_sims_all": sims, "reg_sims_few": few}


def pvr_case():
    stacks, mask, mattr, rattr, rmask = phantom.make_stacks(2, (24, 24, 5), 1.1, 2.2, None, 1.0, 11.0, seed=4,
                                                            orientations=("ax", "sag"))
    P = pvr.make_pvr_problem(stacks, mask, mattr, rattr, rmask, (16, 16), (8, 8))
    o = po.OracleReconstruction(P, po.CANON, pvr=True)
    d = pvr.irtkPatchBasedReconstruction(o, P.patches_per_stack, P.min_intensity, P.max_intensity)
    d.reconstruct_iteration(1)
    return {"pvr_patches_per_stack": np.array(P.patches_per_stack, np.int32),
            "pvr_patch_sum": np.float64(P.slices.astype(np.float64).sum()),
            "pvr_recon": o.recon.copy(), "pvr_scale": d.scale.copy(), "pvr_patch_weight": d.patch_weight.copy(),
            "pvr_em": np.array([d.m_sigma_gpu, d.m_mix_gpu, d.m_m_gpu, d.m_mix_s_gpu], np.float32)}


def main():
    out = {}
    out.update(registration_case())
    out.update(pvr_case())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_v2_reg_pvr.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
