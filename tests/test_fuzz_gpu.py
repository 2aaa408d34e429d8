"""Randomised geometry against the oracle on the default kernels (the cell-owned scatter and gather, csrc/svr_cell.inc):
resolutions from 0.6 to 1.4 mm (pixel densities 0.3 .. 1.6: every automatic cell size), large slice motion (oblique
footprints, pixels off the volume), per-slice thickness and in-plane dimensions that differ from slice to slice, a volume
shifted so that footprints hang over its low and high ends, slice-to-volume and patch-to-volume constants.  What the fixed
problems of the other tests do not vary."""
import copy

import numpy as np
import pytest

from fetalreconstruction_amd import phantom
from tests.util import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    n_stacks = int(rng.integers(2, 4))
    orient = tuple(rng.choice(["ax", "cor", "sag", "ax30"], n_stacks, replace=True))
    res = float(rng.uniform(0.6, 1.4))
    P = phantom.make_problem(n_stacks, (int(rng.integers(16, 30)), int(rng.integers(16, 30)), int(rng.integers(3, 7))),
                             float(rng.uniform(0.8, 1.3)), float(rng.uniform(1.8, 3.0)), None, res, float(rng.uniform(9.0, 13.0)),
                             motion_frac=0.6, motion_mm=3.0, motion_deg=8.0, seed=seed, orientations=orient, name=f"fuzz{seed}")
    P = copy.copy(P)
    P.slice_dim = P.slice_dim.copy()
    P.slice_dim[:, 2] *= rng.uniform(0.7, 1.9, P.ns).astype(np.float32)        # a thickness of its own for every slice
    P.slice_dim[:, 0] *= rng.uniform(0.9, 1.3, P.ns).astype(np.float32)
    P.slice_dim[:, 1] *= rng.uniform(0.9, 1.3, P.ns).astype(np.float32)
    # the volume's frame moved by a few voxels: footprints beyond both ends (negative coordinates alias to 0, RC.cu:382 / 508)
    sh = np.eye(4, dtype=np.float64)
    sh[:3, 3] = rng.uniform(-4.0, 4.0, 3) * res
    i2w = np.asarray(P.recon_i2w, np.float64).reshape(4, 4)
    P.recon_i2w = (sh @ i2w).astype(np.float32).reshape(16)
    P.recon_w2i = np.linalg.inv(sh @ i2w).astype(np.float32).reshape(16)
    return P, rng


@pytest.mark.parametrize("pvr", [False, True])
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SVR_FUZZ_SEEDS", "6"))))      # SVR_FUZZ_SEEDS=60: a longer hunt
def test_random_geometry_against_the_oracle(seed, pvr, oracle_mod):
    from fetalreconstruction_amd import engine as E
    P, rng = _case(seed)
    rec = E.Reconstruction(0)
    if pvr:
        rec.set_option("pvr", 1)
        E.sync_gpu(rec, P, quality_factor=1.0)
    else:
        E.sync_gpu(rec, P)
    orc = oracle_mod.OracleReconstruction(P, oracle_mod.CANON, pvr=pvr)
    assert rec.get_option("back_mode") == 5 and rec.get_option("fwd_mode") == 2
    ones = np.ones(P.ns, np.float32)
    for r in (rec, orc):
        r.UpdateScaleVector(ones, ones)
        r.InitializeEMValues()
        r.GaussianReconstruction()
    ps = rec.debug_get(E.BUF_PSF_SUMS)
    assert np.array_equal(ps != 0, orc.psf_sums != 0)
    assert np.allclose(ps, orc.psf_sums, rtol=1e-6, atol=0, equal_nan=True)
    vw = rec.getVolWeights()
    assert np.array_equal(vw > 0, orc.volw > 0) and rel_err(vw, orc.volw) < TOL
    rec.debug_set(E.BUF_RECONSTRUCTED, orc.recon)
    rec.SimulateSlices(); orc.SimulateSlices()
    assert np.array_equal(rec.debug_get(E.BUF_SIMINSIDE), orc.siminside)
    assert rel_err(rec.debug_get(E.BUF_SIMSLICES), orc.simslices) < TOL and rel_err(rec.debug_get(E.BUF_SIMWEIGHTS), orc.simweights) < TOL
    orc.weights[...] = np.where(orc.slices != -1, rng.uniform(0.0, 1.0, orc.slices.shape) * (rng.uniform(0, 1, orc.slices.shape) > 0.2), 0).astype(np.float32)
    orc.simslices[...] = np.where(orc.slices > 0, orc.slices * rng.uniform(0.8, 1.2, orc.slices.shape), 0).astype(np.float32)
    rec.debug_set(E.BUF_WEIGHTS, orc.weights)
    rec.debug_set(E.BUF_SIMSLICES, orc.simslices)
    sw = rng.uniform(0.0, 1.0, P.ns).astype(np.float32)
    sw[rng.integers(0, P.ns)] = 0.0
    rec.SuperresolutionBackproject(sw); orc.SuperresolutionBackproject(sw)
    cm, ad = rec.debug_get(E.BUF_CONFIDENCE_MAP), rec.debug_get(E.BUF_ADDON)
    assert (orc.cmap > 0).sum() > 100
    assert np.array_equal(cm > 0, orc.cmap > 0)
    assert rel_err(cm, orc.cmap) < TOL and rel_err(ad, orc.addon) < TOL
    if not pvr:
        # round 6: the three forms of the two passes give the SAME BITS on every geometry -- the gather above evaluated and wrote the coefficient
        # table (coeff_lazy), the scatter above streamed it; once more with both streaming it, then with every tap evaluated in both
        assert rec.get_option("coeff_table") == 1 and rec.get_option("coeff_valid") == 1
        cm, ad = cm.copy(), ad.copy()
        rec.debug_set(E.BUF_SIMSLICES, np.zeros_like(orc.simslices))
        rec.SimulateSlices()
        ref = [rec.debug_get(b).copy() for b in (E.BUF_SIMSLICES, E.BUF_SIMWEIGHTS, E.BUF_SIMINSIDE)]
        rec.set_option("coeff_table", 0)
        rec.debug_set(E.BUF_SIMSLICES, np.zeros_like(orc.simslices))
        rec.SimulateSlices()
        for a, b in zip(ref, (E.BUF_SIMSLICES, E.BUF_SIMWEIGHTS, E.BUF_SIMINSIDE)):
            assert np.array_equal(a, rec.debug_get(b), equal_nan=True)
        rec.debug_set(E.BUF_SIMSLICES, orc.simslices)
        rec.SuperresolutionBackproject(sw)
        assert np.array_equal(cm, rec.debug_get(E.BUF_CONFIDENCE_MAP)) and np.array_equal(ad, rec.debug_get(E.BUF_ADDON))
    rec.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SVR_FUZZ_HOST_SEEDS", "4"))))                 # SVR_FUZZ_HOST_SEEDS=40: a longer hunt
def test_cpp_host_on_random_geometry_matches_the_python_driver(seed):
    """svr::irtkReconstruction (one wait per SR iteration: deferred vectors, M-step + E-step fused) against the Python mirror
    of the operator surface (one wait per method) on a second engine: an outer iteration of three SR iterations."""
    from fetalreconstruction_amd import engine as E, host
    from tests.twins.reconstruction import irtkReconstruction
    P, _ = _case(seed)
    kw = dict(max_intensity=P.max_intensity, min_intensity=P.min_intensity)
    ra, rb = E.Reconstruction(0), E.Reconstruction(0)
    E.sync_gpu(ra, P); E.sync_gpu(rb, P)
    hc = host.irtkReconstruction(ra, P.ns, **kw)
    dp = irtkReconstruction(rb, P.ns, **kw)
    for d in (hc, dp):
        d.SetSmoothingParameters(150, 0.02)
        d.reconstruct_iteration(3)
    st = hc.state()
    # (a case whose EM degenerates -- NaN sigma from a volume frame that misses most of the stacks -- must degenerate alike)
    assert np.allclose(st["scale"], dp._scale_gpu, rtol=2e-5, equal_nan=True)
    assert np.allclose(st["slice_weight"], dp._slice_weight_gpu, atol=2e-4, equal_nan=True)
    assert np.allclose([st["sigma"], st["mix"], st["m"], st["mix_s"]], [dp._sigma_gpu, dp._mix_gpu, dp._m_gpu, dp._mix_s_gpu], rtol=2e-5, equal_nan=True)
    assert np.array_equal(st["slice_inside"].astype(bool), np.asarray(dp._slice_inside_gpu, bool))
    va, vb = ra.syncCPU(), rb.syncCPU()
    assert np.array_equal(np.isnan(va), np.isnan(vb)) and rel_err(np.nan_to_num(va), np.nan_to_num(vb)) < 2e-5
    ra.close(); rb.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SVR_FUZZ_PVR_SEEDS", "3"))))                  # SVR_FUZZ_PVR_SEEDS=20: a longer hunt
def test_cpp_pvr_host_on_random_stacks_matches_the_python_loop(seed):
    """svr::irtkPatchBasedReconstruction (csrc/pvr_host.cpp: the patch-level EM on the device, csrc/svr_em.inc patch form -- float Gaussian,
    the potentials through the reference's offset-less copy) against the Python mirror of the loop (tests/twins/pvr.py: the EM on the host,
    numpy float32) on random stacks cut into patches: two to three stacks of different sizes and orientations, so the stacks hold different
    numbers of patches and most patches read another patch's potential; an outer iteration of three SR iterations."""
    from fetalreconstruction_amd import engine as E, host
    from tests.twins import pvr
    rng = np.random.default_rng(7000 + seed)
    n_stacks = int(rng.integers(2, 4))
    orient = tuple(rng.choice(["ax", "cor", "sag"], n_stacks, replace=True))
    stacks, mask, mattr, rattr, rmask = phantom.make_stacks(n_stacks, (int(rng.integers(20, 30)), int(rng.integers(20, 30)), int(rng.integers(3, 6))),
                                                            float(rng.uniform(0.9, 1.3)), float(rng.uniform(1.8, 2.6)), None, 1.0,
                                                            float(rng.uniform(10.0, 12.0)), seed=100 + seed, orientations=orient)
    P = pvr.make_pvr_problem(stacks, mask, mattr, rattr, rmask, (16, 16), (8, 8))
    out = []
    for make in (lambda r: pvr.irtkPatchBasedReconstruction(r, P.patches_per_stack, P.min_intensity, P.max_intensity),
                 lambda r: host.irtkPatchBasedReconstruction(r, P.patches_per_stack, P.min_intensity, P.max_intensity)):
        rec = E.Reconstruction(0)
        rec.set_option("pvr", 1)
        E.sync_gpu(rec, P, quality_factor=1.0)
        d = make(rec)
        d.reconstruct_iteration(3)
        st = d.state() if hasattr(d, "state") else dict(scale=d.scale, patch_weight=d.patch_weight, patch_potential=d.patch_potential,
                                                        **{k: float(getattr(d, k)) for k in ("m_sigma_gpu", "m_mix_gpu", "m_m_gpu", "m_mean_s_gpu",
                                                                                            "m_mean_s2_gpu", "m_sigma_s_gpu", "m_sigma_s2_gpu", "m_mix_s_gpu")})
        out.append((st, rec.syncCPU().copy()))
        rec.close()
    (a, va), (b, vb) = out
    for k in ("m_sigma_gpu", "m_mix_gpu", "m_m_gpu", "m_mean_s_gpu", "m_mean_s2_gpu", "m_sigma_s_gpu", "m_sigma_s2_gpu", "m_mix_s_gpu"):
        assert a[k] == pytest.approx(b[k], rel=2e-4, nan_ok=True), k
    assert np.allclose(a["scale"], b["scale"], rtol=2e-5, equal_nan=True)
    assert np.array_equal(np.asarray(a["patch_potential"]) == -1, np.asarray(b["patch_potential"]) == -1)
    assert np.allclose(a["patch_weight"], b["patch_weight"], atol=2e-4, equal_nan=True)      # expf of the device vs numpy's float32 exp
    assert np.allclose(a["patch_potential"], b["patch_potential"], rtol=1e-5, atol=1e-6, equal_nan=True)
    assert np.array_equal(np.isnan(va), np.isnan(vb)) and rel_err(np.nan_to_num(vb), np.nan_to_num(va)) < 1e-4
