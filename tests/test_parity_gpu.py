"""GPU parity: the HIP engine (through the C-ABI) against the CPU oracle on the same seeded inputs.

Tolerances (float32 path, stated once):
  * index work -- centre voxel, tap -> voxel mapping, epsilon-skip keep masks, hit sets
    (psf_sums != 0, voxcount, siminside, cmap > 0, volw > 0): BIT-EXACT.
  * v_PSF_sums: canonical double sum rounded once on both sides: rel 1e-6.
  * scatter/gather sums (float atomics / wave partial sums vs the oracle's double accumulation):
    |x - ref| <= 2e-5 * max|ref| per buffer (observed ~2e-6).
  * EM scalars (double-accumulated on both sides): rel 1e-5.
"""
import os

import numpy as np
import pytest

from fetalreconstruction_amd import phantom
from tests.twins.reconstruction import irtkReconstruction
from tests.util import rel_err, run_to_state

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_v1.npz")
TOL_SUM = 2e-5
TOL_SCALAR = 1e-5
# scatter variants: 5 = cell-owned LDS planes, staged and combined in a fixed order, no atomics (csrc/svr_cell.inc); 4 =
# wave-owned LDS planes per slice tile, flushed with float atomics; 3 = the workgroup kernel for every tile; 1 = LDS tiles
# with ds_add_f32; 0 = direct atomics
BACK_MODES = [5, 4, 3, 1, 0]


def _engine(prob):
    from fetalreconstruction_amd import engine
    rec = engine.Reconstruction(0)
    engine.sync_gpu(rec, prob)
    return rec


def _drivers(prob, oracle_mod):
    from fetalreconstruction_amd import engine as E
    rec = _engine(prob)
    orc = oracle_mod.OracleReconstruction(prob, oracle_mod.CANON)
    dg = irtkReconstruction(rec, prob.ns, max_intensity=prob.max_intensity, min_intensity=prob.min_intensity)
    do = irtkReconstruction(orc, prob.ns, max_intensity=prob.max_intensity, min_intensity=prob.min_intensity)
    for d in (dg, do):
        d.SetSmoothingParameters(150, 0.02)
    return E, rec, orc, dg, do


def test_native_library_is_loaded():
    from fetalreconstruction_amd import engine
    engine.Reconstruction(0).close()
    maps = open("/proc/self/maps").read()
    assert "libsvr_hip.so" in maps


def test_psf_taps_are_bit_identical(tiny, oracle_mod, golden=True):
    """The canonical PSF sequence on the device against the oracle: every one of the 4096 tap
    values, the epsilon-skip keep mask and the centre voxel, bit for bit; and the keep masks
    against the committed golden census."""
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    rec.UpdateScaleVector(np.ones(tiny.ns), np.ones(tiny.ns))
    act = np.argwhere(tiny.slices != -1)
    rng = np.random.default_rng(5)
    for i in rng.choice(len(act), 150, replace=False):
        sl, py, px = act[i]
        v, c = rec.probe_pixel(sl, px, py)
        n, bits, vals, cc = orc.tap_census(sl, px, py, with_vals=True)
        assert np.array_equal(c, cc.astype(np.int32))
        assert np.array_equal(v.view(np.uint32), vals.view(np.uint32))     # values and skips (-1)
    if not golden:
        return
    gold = np.load(GOLD)
    for p, bits in zip(gold["census_pix"], gold["census_bits"]):
        v, _ = rec.probe_pixel(p[0], p[2], p[1])
        kept = ~(v < 0)
        packed = np.packbits(kept, bitorder="little").view(np.uint64)
        assert np.array_equal(packed, bits)


@pytest.mark.parametrize("thickness", [1.0, 0.3])
def test_thin_slices_take_the_exponential_per_tap(oracle_mod, thickness):
    """The Gaussian factor of a row is stepped outwards from its two central taps (gauss_pairs / canon_gauss_row); rows whose
    central factors are too small to start from (thickness 1.0: thin slices seen obliquely) and slices whose first ratio
    could overflow (thickness 0.3: w < 0) evaluate the exponential per tap instead.  Same rows on both sides, bit for bit,
    and the kernels that consume them agree."""
    from fetalreconstruction_amd import phantom
    prob = phantom.make_problem(3, (24, 24, 6), 1.1, 2.2, thickness, 1.0, 11.0, seed=3, orientations=("ax", "cor", "sag"), name="thin")
    E, rec, orc, dg, do = _drivers(prob, oracle_mod)
    rec.UpdateScaleVector(np.ones(prob.ns), np.ones(prob.ns))
    act = np.argwhere(prob.slices != -1)
    rng = np.random.default_rng(6)
    direct_rows = 0
    for i in rng.choice(len(act), 60, replace=False):
        sl, py, px = act[i]
        v, c = rec.probe_pixel(sl, px, py)
        n, bits, vals, cc = orc.tap_census(sl, px, py, with_vals=True)
        assert np.array_equal(c, cc.astype(np.int32))
        assert np.array_equal(v.view(np.uint32), vals.view(np.uint32))
        raw = orc.psf_values(sl, px, py).reshape(16, 16, 16)
        direct_rows += int(((raw[:, :, 7] == 0) & (raw[:, :, 8] == 0)).sum())          # central factors flushed to 0: rows that start too far out
    assert direct_rows > 0
    for d in (dg, do):
        d.InitializeEMValuesGPU()
        d.GaussianReconstructionGPU()
        d.SimulateSlicesGPU()
    assert rel_err(rec.syncCPU(), orc.recon) < TOL_SUM
    assert rel_err(rec.debug_get(E.BUF_SIMSLICES), orc.simslices) < TOL_SUM


@pytest.mark.parametrize("gauss_mode,fwd_mode", [(1, 2), (1, 1), (0, 1), (0, 0)])
def test_gaussian_reconstruction_parity(tiny, oracle_mod, gauss_mode, fwd_mode):
    """gauss_mode 1 = unit-based pass 1 (dead-unit shortcut; over (cell, plane) items with fwd_mode 2, per slice tile with fwd_mode 1)
    + the LDS scatter, 0 = wave-per-pixel kernel with atomics."""
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    rec.set_option("gauss_mode", gauss_mode)
    rec.set_option("legacy_kernels", 1); rec.set_option("fwd_mode", fwd_mode)
    run_to_state(dg, "gauss")
    run_to_state(do, "gauss")
    ps = rec.debug_get(E.BUF_PSF_SUMS)
    assert np.array_equal(ps != 0, orc.psf_sums != 0)                       # hit set: exact
    assert rel_err(ps, orc.psf_sums) < 1e-6
    assert np.array_equal(rec.debug_get(E.BUF_VOXEL_COUNT), orc.voxcount)
    vw = rec.getVolWeights()
    assert np.array_equal(vw > 0, orc.volw > 0)
    assert rel_err(vw, orc.volw) < TOL_SUM
    assert rel_err(rec.syncCPU(), orc.recon) < TOL_SUM


def test_against_committed_golden(tiny, oracle_mod):
    """Same checks against tests/golden/tiny_v1.npz (no oracle needed at run time)."""
    gold = np.load(GOLD)
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    run_to_state(dg, "gauss")
    assert np.array_equal(rec.debug_get(E.BUF_PSF_SUMS) != 0, gold["psf_sums"] != 0)
    assert rel_err(rec.debug_get(E.BUF_PSF_SUMS), gold["psf_sums"]) < 1e-6
    assert rel_err(rec.syncCPU(), gold["gauss_recon"]) < TOL_SUM
    dg.SimulateSlicesGPU()
    assert rel_err(rec.debug_get(E.BUF_SIMSLICES), gold["simslices0"]) < TOL_SUM
    assert np.array_equal(rec.debug_get(E.BUF_SIMINSIDE), gold["siminside0"])


@pytest.mark.parametrize("fwd_mode", [2, 1, 0])
def test_forward_projection_parity(tiny, oracle_mod, fwd_mode):
    """fwd_mode 2 = the gather over the (cell, plane) items of the scatter without atomics (csrc/svr_cell.inc), 1 = unit-based
    gather per slice tile (float2 {V m, m} box, dead-unit shortcut), 0 = wave-per-pixel kernel."""
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    rec.set_option("legacy_kernels", 1); rec.set_option("fwd_mode", fwd_mode)
    run_to_state(dg, "sim")
    run_to_state(do, "sim")
    assert np.array_equal(dg._slice_inside_gpu, do._slice_inside_gpu)
    assert np.array_equal(rec.debug_get(E.BUF_SIMINSIDE), orc.siminside)
    assert rel_err(rec.debug_get(E.BUF_SIMWEIGHTS), orc.simweights) < TOL_SUM
    assert rel_err(rec.debug_get(E.BUF_SIMSLICES), orc.simslices) < TOL_SUM


def test_gather_in_pieces(tiny, oracle_mod, monkeypatch):
    """A dispatch holds 2^32 - 1 work-items, i.e. 8.4 M tiles of the gather's 512-lane workgroups; longer tile lists (2 x 2 tiles
    of a 0.5 mm patch-based case: 12.2 M) go out in pieces.  SVR_FWD_PIECE shortens the pieces so that the tiny problem takes
    the same path: Gaussian pass 1 and the simulated slices must not change."""
    outs = []
    for piece in (None, "48"):
        if piece:
            monkeypatch.setenv("SVR_FWD_PIECE", piece)
        E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
        run_to_state(dg, "sim")
        outs.append((rec.debug_get(E.BUF_PSF_SUMS), rec.syncCPU(), rec.debug_get(E.BUF_SIMSLICES), rec.debug_get(E.BUF_SIMWEIGHTS),
                     rec.debug_get(E.BUF_SIMINSIDE)))
    assert rec.counters()["Va"] // 32 > 3 * 48                               # several pieces even with the largest tiles (8 x 4)
    for k, (a, b) in enumerate(zip(*outs)):
        if k in (1, 2):                                                       # the volume comes out of float atomics (pass 2), the simulation reads it
            assert rel_err(a, b) < TOL_SUM
        else:
            assert np.array_equal(a, b)
    run_to_state(do, "sim")
    assert rel_err(outs[1][2], orc.simslices) < TOL_SUM


@pytest.mark.parametrize("back_mode", BACK_MODES)
def test_backprojection_parity(tiny, oracle_mod, back_mode):
    """back_mode 4 = wave-owned LDS planes with the dead-unit shortcut (default), 3 = the workgroup kernel for every tile,
    1 = LDS tiles with ds_add_f32, 0 = direct atomics."""
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    rec.set_option("legacy_kernels", 1); rec.set_option("back_mode", back_mode)
    run_to_state(dg, "scale")
    run_to_state(do, "scale")
    # identical inputs for the scatter: copy the oracle's per-pixel state onto the device
    rec.debug_set(E.BUF_WEIGHTS, orc.weights)
    rec.debug_set(E.BUF_SIMSLICES, orc.simslices)
    rec.debug_set(E.BUF_PSF_SUMS, orc.psf_sums)
    rec.UpdateScaleVector(orc.d_scales, orc.slice_weights)
    rec.SuperresolutionBackproject(orc.slice_weights)
    orc.SuperresolutionBackproject(orc.slice_weights)
    cm, ad = rec.debug_get(E.BUF_CONFIDENCE_MAP), rec.debug_get(E.BUF_ADDON)
    assert np.array_equal(cm > 0, orc.cmap > 0)                           # footprint: exact
    assert rel_err(cm, orc.cmap) < TOL_SUM
    assert rel_err(ad, orc.addon) < TOL_SUM


@pytest.mark.parametrize("reg_mode", [1, 0])
@pytest.mark.parametrize("adaptive", [False, True])
def test_regulariser_parity(tiny, oracle_mod, reg_mode, adaptive):
    """reg_mode 1 = k_regul_fused (csrc/svr_regul.inc: Prep + regulariser in one kernel, LDS planes, float32 weights through
    v_rsq_f32), 0 = k_reg_prep + k_regularize (the reference's operations: fp64 sqrt and division per neighbour pair).  Row a7's
    tolerance: 1e-6 of the volume's maximum against the oracle's double-precision weights, for both."""
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    rec.set_option("reg_mode", reg_mode)
    run_to_state(do, "scale")
    orc.SuperresolutionBackproject(orc.slice_weights)
    if adaptive:                                                      # confidences other than {0, 1} reach the regulariser only then
        assert len(np.unique(orc.cmap)) > 100
    rec.debug_set(E.BUF_RECONSTRUCTED, orc.recon)
    rec.debug_set(E.BUF_ADDON, orc.addon)
    rec.debug_set(E.BUF_CONFIDENCE_MAP, orc.cmap)
    args = (adaptive, do._alpha, do._min_intensity, do._max_intensity, do._delta, do._lambda)
    rec.SuperresolutionUpdate(*args)
    orc.SuperresolutionUpdate(*args)
    assert rel_err(rec.syncCPU(), orc.recon) < 1e-6
    assert rel_err(rec.debug_get(E.BUF_ADDON), orc.addon) < 1e-6      # Prep's own changes of the two buffers (settled for the reader in mode 1)
    assert np.array_equal(rec.debug_get(E.BUF_CONFIDENCE_MAP), orc.cmap)
    # a second update on top of the first (the volume lives in the other buffer now), strong smoothing, clamped range
    args2 = (adaptive, do._alpha, float(np.percentile(orc.recon, 30)), float(np.percentile(orc.recon, 90)), 40.0, 0.5 * 40.0 ** 2 * 0.06 / do._alpha)
    rec.debug_set(E.BUF_ADDON, orc.addon); rec.debug_set(E.BUF_CONFIDENCE_MAP, orc.cmap)
    rec.SuperresolutionUpdate(*args2)
    orc.SuperresolutionUpdate(*args2)
    assert rel_err(rec.syncCPU(), orc.recon) < 1e-6


def test_fused_update_skips_nothing_it_should_not(tiny, oracle_mod):
    """With cmap straight from the scatter the fused update writes zeros for tiles outside the mask's box without loading
    anything; the same update with the shortcut off (cmap handed in through debug_set), the two-kernel form and the oracle
    give the same volume, through two SR iterations of the whole loop, on a volume whose mask sits in one corner."""
    import copy
    from fetalreconstruction_amd import engine as E
    P = copy.copy(tiny)
    P.mask = tiny.mask.copy()
    P.mask[:, :, : tiny.mask.shape[2] // 2] = 0                      # (z, y, x): the mask's box ends mid-volume
    P.mask[: tiny.mask.shape[0] // 3] = 0
    outs = []
    for reg_mode in (1, 0):
        rec = E.Reconstruction(0)
        E.sync_gpu(rec, P)
        rec.set_option("reg_mode", reg_mode)
        d = irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
        d.SetSmoothingParameters(150, 0.02)
        d.reconstruct_iteration(2)
        outs.append(rec.syncCPU())
    orc = oracle_mod.OracleReconstruction(P, oracle_mod.CANON)
    do = irtkReconstruction(orc, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
    do.SetSmoothingParameters(150, 0.02)
    do.reconstruct_iteration(2)
    assert rel_err(outs[0], outs[1]) < 2e-6
    assert rel_err(outs[0], orc.recon) < 1e-4                            # whole iterations, each side evolving its own state
    assert np.array_equal(outs[0] != 0, outs[1] != 0)


def test_em_steps_parity(tiny, oracle_mod):
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    run_to_state(do, "sim")
    run_to_state(dg, "sim")
    for b, a in ((E.BUF_SIMSLICES, orc.simslices), (E.BUF_SIMWEIGHTS, orc.simweights), (E.BUF_SIMINSIDE, orc.siminside)):
        rec.debug_set(b, a)
    sg, so = rec.InitializeRobustStatistics(), orc.InitializeRobustStatistics()
    assert abs(sg - so) <= TOL_SCALAR * abs(so)
    m = 1.0 / (2.1 * tiny.max_intensity - 1.9 * tiny.min_intensity)
    pg, po_ = rec.EStep(m, so, 0.9), orc.EStep(m, so, 0.9)
    assert np.array_equal(pg < 0, po_ < 0)
    assert rel_err(pg, po_) < TOL_SCALAR
    assert rel_err(rec.debug_get(E.BUF_WEIGHTS), orc.weights, floor=1.0) < 1e-5
    rec.debug_set(E.BUF_WEIGHTS, orc.weights)
    assert rel_err(rec.CalculateScaleVector(), orc.CalculateScaleVector()) < TOL_SCALAR
    mg, mo = rec.MStepSums(), orc.MStepSums()
    assert np.allclose(mg, mo, rtol=TOL_SCALAR, atol=0)
    a, b = rec.MStep(2, 1e-4, so, 0.9), orc.MStep(2, 1e-4, so, 0.9)
    assert np.allclose(a, b, rtol=TOL_SCALAR)


@pytest.mark.parametrize("modes", [(5, 2), (4, 1), (3, 0)])
def test_slices_of_different_thickness_side_by_side_in_the_cell_lists(tiny, oracle_mod, modes):
    """The cell lists put pixels of different slices next to each other (sorted by cell first), and the records' dead-unit
    bits are worked out one pixel per thread: every thread needs ITS slice's constants.  (Round 3 read inv2s2 / invD through
    readfirstlane there -- the first lane's slice decided for all 64 -- which no test saw while all slices had one thickness:
    a unit wrongly declared dead drops its taps.)  Every other slice gets 1.8 x the thickness: scatter and gather on the
    cells against the oracle, and the table scatter (which reads what k_coeff_build stored for the units IT found live) bit for
    bit against the on-the-fly scatter."""
    import copy
    from fetalreconstruction_amd import engine as E
    P = copy.copy(tiny)
    P.slice_dim = tiny.slice_dim.copy()
    P.slice_dim[1::2, 2] *= 1.8
    P.slice_dim[2::3, 0] *= 1.3                                            # ... and every third other in-plane voxel sizes (the PSF's kx, ky)
    P.slice_dim[2::3, 1] *= 1.2
    orc = oracle_mod.OracleReconstruction(P, oracle_mod.CANON)
    rng = np.random.default_rng(5)
    ones = np.ones(P.ns, np.float32)
    outs = {}
    for tab in (0, 1):
        rec = E.Reconstruction(0)
        E.sync_gpu(rec, P)
        assert rec.get_option("back_mode") == 5 and rec.get_option("fwd_mode") == 2      # the defaults: the cell kernels
        rec.set_option("legacy_kernels", 1); rec.set_option("back_mode", modes[0]); rec.set_option("fwd_mode", modes[1]); rec.set_option("gauss_mode", 1 if modes[0] >= 3 else 0)
        for r in ((rec, orc) if tab == 0 else (rec,)):
            r.UpdateScaleVector(ones, ones)
            r.InitializeEMValues()
            r.GaussianReconstruction()
        assert np.array_equal(rec.debug_get(E.BUF_PSF_SUMS) != 0, orc.psf_sums != 0)
        assert rel_err(rec.getVolWeights(), orc.volw) < TOL_SUM and np.array_equal(rec.getVolWeights() > 0, orc.volw > 0)
        rec.debug_set(E.BUF_RECONSTRUCTED, orc.recon)
        rec.set_option("coeff_table", tab)
        if tab == 0:
            orc.SimulateSlices()
            orc.weights[...] = np.where(orc.slices != -1, rng.uniform(0.2, 1.0, orc.slices.shape), 0).astype(np.float32)
            sim_in = np.where(orc.slices > 0, orc.slices * rng.uniform(0.8, 1.2, orc.slices.shape), 0).astype(np.float32)
        rec.SimulateSlices()
        assert rec.get_option("coeff_table") == tab
        assert np.array_equal(rec.debug_get(E.BUF_SIMINSIDE), orc.siminside) and rel_err(rec.debug_get(E.BUF_SIMSLICES), orc.simslices) < TOL_SUM
        rec.debug_set(E.BUF_SIMSLICES, sim_in)
        rec.debug_set(E.BUF_WEIGHTS, orc.weights)
        rec.SuperresolutionBackproject(ones)
        outs[tab] = (rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy())
        rec.close()
    orc.simslices[...] = sim_in
    orc.SuperresolutionBackproject(ones)
    assert np.array_equal(outs[0][1] > 0, orc.cmap > 0)
    assert rel_err(outs[0][1], orc.cmap) < TOL_SUM and rel_err(outs[0][0], orc.addon) < TOL_SUM
    if modes[0] == 5:                                                      # no atomics: table and on-the-fly scatter give the same bits
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    else:
        assert rel_err(outs[1][1], orc.cmap) < TOL_SUM and rel_err(outs[1][0], orc.addon) < TOL_SUM


@pytest.mark.parametrize("pvr", [False, True])
def test_deferred_reads_and_the_fused_m_e_step_give_the_separate_calls_bits(tiny, pvr):
    """svr_calculate_scale_vector / svr_simulate_slices with NULL (nothing comes back, no wait), svr_get_scale_vector,
    svr_get_slice_inside, svr_adopt_scale_vector and svr_mstep_estep (the M-step's scalars worked out on the device) against
    CalculateScaleVector -> SimulateSlices -> MStep -> EStep one after the other: every number bit for bit, over three rounds
    (the scale vector lags one call behind on the device, RC.cu:3238)."""
    from fetalreconstruction_amd import engine as E

    def fresh():
        rec = E.Reconstruction(0)
        if pvr:
            rec.set_option("pvr", 1)
            E.sync_gpu(rec, tiny, quality_factor=1.0)
        else:
            E.sync_gpu(rec, tiny)
        ones = np.ones(tiny.ns, np.float32)
        rec.UpdateScaleVector(ones, ones)
        rec.InitializeEMValues()
        rec.GaussianReconstruction()
        rec.SimulateSlices()
        return rec

    a, b = fresh(), fresh()
    for r in (a, b):
        r.set_option("legacy_kernels", 1); r.set_option("back_mode", 5)                                       # the scatter without atomics: the two engines stay bit-identical
    for buf in (E.BUF_RECONSTRUCTED, E.BUF_VOL_WEIGHTS, E.BUF_SIMSLICES, E.BUF_SIMWEIGHTS, E.BUF_SIMINSIDE, E.BUF_PSF_SUMS):
        b.debug_set(buf, a.debug_get(buf))                                 # (the Gaussian pass of the patch-based path adds with atomics)
    sigma = a.InitializeRobustStatistics()
    assert sigma == b.InitializeRobustStatistics()
    m = 1.0 / (2.1 * tiny.max_intensity - 1.9 * tiny.min_intensity)
    pa, pb = a.EStep(m, sigma, 0.9), b.EStep(m, sigma, 0.9)
    assert np.array_equal(pa, pb)
    sa, mixa, sb, mixb = sigma, 0.9, sigma, 0.9
    w = np.ones(tiny.ns, np.float32)
    for it in range(1, 4):
        sc_a = a.CalculateScaleVector()
        b.CalculateScaleVectorDeferred()
        if pvr:
            a.UpdateScaleVector(sc_a, w)                                   # copyToScales
            b.AdoptScaleVector()
        for r in (a, b):
            r.Superresolution(it, w, False, 2.5, tiny.min_intensity, tiny.max_intensity, 150.0, 450.0)
        in_a = a.SimulateSlices()
        b.SimulateSlicesDeferred()
        s5_b, sc_f, in_f = b.MStepSumsFetch(want_scale=True, want_inside=True)   # the sharded hosts' form: sums + deferred vectors
        assert np.array_equal(s5_b, a.MStepSums()) and np.array_equal(sc_f, sc_a) and np.array_equal(in_f, in_a)
        sa, mixa, ma = a.MStep(it, 1e-4, sa, mixa)
        pa = a.EStep(ma, sa, mixa)
        sb, mixb, mb, pb, sc_b, in_b = b.MStepEStep(it, 1e-4, sb, mixb, want_scale=True, want_inside=True)
        if pvr:                                                            # the patch-based M-step has no FLT_MAX / FLT_MIN clamps: same numbers here
            assert np.isfinite(mb)
        assert (sa, mixa, ma) == (sb, mixb, mb), (it, sa, mixa, ma, sb, mixb, mb)
        assert np.array_equal(pa, pb) and np.array_equal(sc_a, sc_b) and np.array_equal(in_a, in_b)
        assert np.array_equal(sc_a, b.GetScaleVector()) and np.array_equal(in_a, b.GetSliceInside())
        assert np.array_equal(a.debug_get(E.BUF_WEIGHTS), b.debug_get(E.BUF_WEIGHTS))
        assert np.array_equal(a.syncCPU(), b.syncCPU())
    a.close(); b.close()


def test_full_iteration_tracks_the_oracle(tiny, oracle_mod):
    """Gaussian init + 2 SR iterations end to end, each side on its own state."""
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    dg.reconstruct_iteration(2)
    do.reconstruct_iteration(2)
    assert np.allclose(dg._scale_gpu, do._scale_gpu, rtol=1e-4)
    assert np.allclose(dg._slice_weight_gpu, do._slice_weight_gpu, atol=1e-3)
    assert np.allclose([dg._sigma_gpu, dg._mix_gpu, dg._m_gpu], [do._sigma_gpu, do._mix_gpu, do._m_gpu], rtol=1e-4)
    g, o = rec.syncCPU(), orc.recon
    assert np.array_equal(g == -1, o == -1)
    assert rel_err(g, o) < 1e-4


@pytest.mark.parametrize("shift", [(-14.2, -14.4, -14.6), (14.3, 14.1, 13.9), (-14.2, 14.1, 0.3)])
@pytest.mark.parametrize("back_mode", [5, 4, 3, 1])
def test_volume_boundary_quirks_on_device(oracle_mod, shift, back_mode):
    """Slices hanging off the volume: negative coordinates alias to index 0 (float->uint
    saturation), taps beyond the high end are dropped -- in the Gaussian scatter, the forward
    gather and both tiled back-projections, exactly like the oracle."""
    from fetalreconstruction_amd import geometry as geo
    P = phantom.make_problem(1, (12, 12, 2), 1.0, 2.0, None, 1.0, 14.0, seed=5, orientations=("ax",),
                             motion_frac=0.0, noise_sigma=0.0)
    t = geo.rigid_matrix(tx=shift[0], ty=shift[1], tz=shift[2])
    for k in range(P.ns):
        P.slice_t[k] = geo.to_matrix4(t)
        P.slice_tinv[k] = geo.to_matrix4(np.linalg.inv(t))
    P.mask[...] = 1.0
    P.slices[...] = 100.0
    P.slices[:, ::3, ::2] = 140.0
    E, rec, orc, dg, do = _drivers(P, oracle_mod)
    rec.set_option("legacy_kernels", 1); rec.set_option("back_mode", back_mode)
    rec.set_option("legacy_kernels", 1); rec.set_option("fwd_mode", 1 if back_mode >= 3 else 0)
    rec.set_option("gauss_mode", 1 if back_mode >= 3 else 0)
    run_to_state(dg, "sim")
    run_to_state(do, "sim")
    assert (orc.psf_sums != 0).any()
    assert np.array_equal(rec.debug_get(E.BUF_PSF_SUMS) != 0, orc.psf_sums != 0)
    assert rel_err(rec.getVolWeights(), orc.volw) < TOL_SUM
    assert rel_err(rec.debug_get(E.BUF_SIMSLICES), orc.simslices) < TOL_SUM
    for e in (rec, orc):
        e.InitializeEMValues()
    orc.simslices[...] = np.where(orc.simslices > 0, orc.simslices * 0.9, 0)   # a non-zero residual
    rec.debug_set(E.BUF_SIMSLICES, orc.simslices)
    rec.debug_set(E.BUF_PSF_SUMS, orc.psf_sums)
    rec.SuperresolutionBackproject(np.ones(P.ns, np.float32))
    orc.SuperresolutionBackproject(np.ones(P.ns, np.float32))
    cm = rec.debug_get(E.BUF_CONFIDENCE_MAP)
    assert (orc.cmap > 0).any()
    assert np.array_equal(cm > 0, orc.cmap > 0)
    assert rel_err(cm, orc.cmap) < TOL_SUM
    assert rel_err(rec.debug_get(E.BUF_ADDON), orc.addon) < TOL_SUM


def test_cpp_host_object_matches_python_driver_and_oracle(tiny, oracle_mod):
    """svr::irtkReconstruction (C++, csrc/svr_host.cpp) against the Python mirror on a second
    engine and against the oracle-driven run: same host state, same volume."""
    from fetalreconstruction_amd import host
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    rec2 = _engine(tiny)
    hc = host.irtkReconstruction(rec2, tiny.ns, max_intensity=tiny.max_intensity, min_intensity=tiny.min_intensity)
    hc.SetSmoothingParameters(150, 0.02)
    hc.reconstruct_iteration(2)
    dg.reconstruct_iteration(2)
    do.reconstruct_iteration(2)
    st = hc.state()
    for ref, tol in ((dg, 2e-5), (do, 1e-4)):
        assert np.allclose(st["scale"], ref._scale_gpu, rtol=tol)
        assert np.allclose(st["slice_weight"], ref._slice_weight_gpu, atol=10 * tol)
        assert np.allclose([st["sigma"], st["mix"], st["m"], st["mix_s"]],
                           [ref._sigma_gpu, ref._mix_gpu, ref._m_gpu, ref._mix_s_gpu], rtol=tol)
    assert np.array_equal(st["slice_potential"] < 0, dg._slice_potential_gpu < 0)
    assert rel_err(rec2.syncCPU(), rec.syncCPU()) < 2e-5
    assert rel_err(rec2.syncCPU(), orc.recon) < 1e-4


def test_cpp_host_in_a_sharded_numbering_gives_the_references_order_its_results(tiny):
    """svrh_set_unit_order (round 5): a launcher that deals the r-th part of EVERY stack to rank r uploads the slices in that order and tells
    the host object; everything per slice is indifferent to the numbering, and the slice-level EM (RG.cc:3282-3420), whose sums run over the
    slices in slice order, keeps running in the reference's order.  The whole problem in the numbering of a 3-rank spatial sharding on one
    engine against the plain order on another: the same volume (up to the order of float additions inside a cell's runs), the same EM scalars,
    and per-slice vectors that are each other's permutation -- force-excluded slices named in the object's numbering."""
    from fetalreconstruction_amd import host, phantom
    from fetalreconstruction_amd.sharding import shard_units
    act = (tiny.slices != -1).reshape(tiny.ns, -1).sum(1)
    order, ranges = shard_units(act, tiny.stack_index, 3, "spatial")
    assert not np.array_equal(order, np.arange(tiny.ns))
    inv = np.argsort(order)
    excluded = [2, tiny.ns - 3]                                  # reference indices
    out = []
    for perm in (None, order):
        P = tiny if perm is None else phantom.sub_problem(tiny, 0, 0, select=perm)
        rec = _engine(P)
        hc = host.irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
        if perm is not None:
            hc.set_unit_order(perm)
        hc.SetForceExcludedSlices(excluded if perm is None else [int(inv[i]) for i in excluded])
        hc.SetSmoothingParameters(150, 0.02)
        hc.reconstruct_iteration(3)
        out.append((rec.syncCPU().copy(), hc.state()))
        rec.close()
    (v0, s0), (v1, s1) = out
    assert rel_err(v1, v0) < 2e-5
    for k in ("sigma", "mix", "m", "mean_s", "mean_s2", "sigma_s", "sigma_s2", "mix_s"):
        assert s1[k] == pytest.approx(s0[k], rel=1e-5), k
    assert np.allclose(s1["scale"], s0["scale"][order], rtol=1e-5) and np.allclose(s1["slice_weight"], s0["slice_weight"][order], atol=1e-4)
    assert np.allclose(s1["slice_potential"], s0["slice_potential"][order], rtol=1e-4, atol=1e-7)
    assert (s0["slice_weight"][excluded] == 0).all() and (s1["slice_weight"][inv[excluded]] == 0).all()
    assert 0 < (s0["slice_weight"] > 0.5).sum() < tiny.ns or (s0["slice_weight"] > 0).sum() == tiny.ns - 2
    # not a permutation: refused
    rec = _engine(tiny)
    hc = host.irtkReconstruction(rec, tiny.ns, max_intensity=tiny.max_intensity, min_intensity=tiny.min_intensity)
    bad = order.copy(); bad[0] = bad[1]
    with pytest.raises(Exception):
        hc.set_unit_order(bad)
    rec.close()


def test_ragged_and_empty_inputs(oracle_mod):
    """Slices of different sizes padded with -1 (RG.cc:269-311), an all-padding slice and a slice
    grid that is not a multiple of any block size."""
    P = phantom.make_problem(2, (37, 29, 3), 1.3, 2.6, None, 1.0, 13.0, seed=9, orientations=("cor", "sag"))
    P.slices[1, 20:, :] = -1
    P.slices[1, :, 25:] = -1          # a smaller slice inside the padded grid
    P.slices[4, :, :] = -1            # an empty slice
    E, rec, orc, dg, do = _drivers(P, oracle_mod)
    dg.reconstruct_iteration(1)
    do.reconstruct_iteration(1)
    assert do._slice_weight_gpu[4] == 0 and dg._slice_weight_gpu[4] == 0
    assert rel_err(rec.syncCPU(), orc.recon) < 1e-4


def test_too_many_pixels_for_one_context_are_refused():
    """Slice-grid indices are 32-bit: 2^31 pixels per context and beyond are refused loudly (below that every launch that takes
    a wavefront per pixel or per tile goes out in pieces, test_lists_go_out_in_pieces)."""
    from fetalreconstruction_amd import engine as E
    rec = E.Reconstruction(0)
    with pytest.raises(E.SvrError, match="2\\^31"):
        rec._ck(rec._lib.svr_init_storage_volumes(rec._h, E._p(np.array([65536, 32768, 1], np.uint32)), E._p(np.ones(3, np.float32))))


@pytest.mark.parametrize("modes", [dict(back_mode=5, fwd_mode=1, gauss_mode=1), dict(back_mode=4, fwd_mode=1, gauss_mode=1),
                                   dict(back_mode=3, fwd_mode=0, gauss_mode=0), dict(back_mode=0, fwd_mode=0, gauss_mode=0)])
def test_lists_go_out_in_pieces(tiny, oracle_mod, monkeypatch, modes):
    """A dispatch holds 2^32 - 1 work-items: pixel lists (a wavefront per pixel), tile lists and the cell scatter's item list are
    launched in pieces (the reference chunks its launches too: MAX_SLICES_PER_RUN, RC.cu:2207-2219, 2414-2432).  SVR_LIST_PIECE
    shortens the pieces so that the tiny problem takes that path in every kernel family -- tile building, the tiled and the
    cell-owned scatter, the wave-per-pixel kernels, the coefficient table -- and must still agree with the oracle."""
    monkeypatch.setenv("SVR_LIST_PIECE", "40")
    monkeypatch.setenv("SVR_FWD_PIECE", "48")
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    rec.set_option("legacy_kernels", 1)                                    # (the generations before the tile fallback are asked for by name here)
    for k, v in modes.items():
        rec.set_option(k, v)
    dg.reconstruct_iteration(2)
    do.reconstruct_iteration(2)
    g, o = rec.syncCPU(), orc.recon
    assert np.array_equal(g == -1, o == -1) and rel_err(g, o) < 1e-4
    assert np.array_equal(rec.debug_get(E.BUF_PSF_SUMS) != 0, orc.psf_sums != 0)
    assert np.array_equal(rec.debug_get(E.BUF_CONFIDENCE_MAP) > 0, orc.cmap > 0)
    if modes["back_mode"] == 4:                                            # ... and the table built and streamed in pieces
        rec.set_option("coeff_table", 1)
        sim0 = rec.debug_get(E.BUF_SIMSLICES).copy()
        rec.SimulateSlices()
        assert rec.get_option("coeff_table") == 1 and np.array_equal(rec.debug_get(E.BUF_SIMSLICES), sim0)


def test_sixteen_stacks_on_one_context():
    """16 stacks of 64 slices of 256^2 pixels = 2^26 slice pixels in ONE context (refused until round 3: a wavefront per pixel of
    such a list does not fit one dispatch).  Too big for the oracle: forward projection and scatter are adjoint,
    <A V, e> = <V, A^T e>, and every active pixel got its simulated value."""
    from fetalreconstruction_amd import engine as E
    P = phantom.make_problem(16, (256, 256, 64), 1.0, 2.5, 2.5, 0.75, 100.0, seed=3, orientations=("ax", "cor", "sag"), name="S16")
    assert P.slices.size == 1 << 26
    rec = _engine(P)
    rec.UpdateScaleVector(np.ones(P.ns), np.ones(P.ns))
    rec.InitializeEMValues()
    rec.GaussianReconstruction()
    rng = np.random.default_rng(0)
    V = rng.uniform(0.5, 1.5, P.nvox).astype(np.float32)
    rec.debug_set(E.BUF_RECONSTRUCTED, V)
    rec.SimulateSlices()
    sim = rec.debug_get(E.BUF_SIMSLICES).astype(np.float64)
    sw = rec.debug_get(E.BUF_SIMWEIGHTS).astype(np.float64)
    ps = rec.debug_get(E.BUF_PSF_SUMS)
    act = (P.slices != -1) & (ps != 0)
    assert act.sum() > 20e6 and (sw[act] > 0).mean() > 0.99
    s = P.slices.astype(np.float32)
    r = rng.uniform(-1, 1, P.slices.shape).astype(np.float32)
    simp = np.where(act, s - r, 0.0).astype(np.float32)
    e = np.where(act & (simp > 0), s.astype(np.float64) - simp.astype(np.float64), 0.0)
    rec.debug_set(E.BUF_SIMSLICES, simp)
    rec.debug_set(E.BUF_WEIGHTS, np.ones(P.slices.shape, np.float32))
    rec.SuperresolutionBackproject(np.ones(P.ns, np.float32))
    addon = rec.debug_get(E.BUF_ADDON).astype(np.float64)
    lhs, rhs = float(np.sum(sim * sw * e)), float(np.sum(addon * V.astype(np.float64)))
    scale = float(np.sum(np.abs(sim * sw * e)))
    assert abs(lhs - rhs) <= 2e-5 * scale, (lhs, rhs, scale)
    assert rec.counters()["Va"] == int(act.sum())


@pytest.mark.parametrize("workload", ["P4", "S8"])
def test_adjointness_at_full_size(workload):
    """Size-independent property on the full-size workloads (too big for the oracle) -- P4 = BASELINE.json configs[1]
    (4 stacks, 1.0 mm), S8 = configs[3] (8 stacks of 64 x 256^2 slices, 0.75 mm, 33.5 M pixels, 40 M voxels): the forward
    projection and the scatter are adjoint, <A V, e> = <V, A^T e> with unit voxel/slice weights."""
    from fetalreconstruction_amd import engine as E, workloads
    P = workloads.get(workload)                                   # the bench's own workloads (P4 = the bundled mask's frame)
    rec = _engine(P)
    rec.UpdateScaleVector(np.ones(P.ns), np.ones(P.ns))
    rec.InitializeEMValues()
    rec.GaussianReconstruction()
    rng = np.random.default_rng(0)
    nv = P.nvox
    V = rng.uniform(0.5, 1.5, nv).astype(np.float32)
    rec.debug_set(E.BUF_RECONSTRUCTED, V)
    rec.SimulateSlices()
    sim = rec.debug_get(E.BUF_SIMSLICES).astype(np.float64)
    sw = rec.debug_get(E.BUF_SIMWEIGHTS).astype(np.float64)      # (A V) = sim * simweight
    ps = rec.debug_get(E.BUF_PSF_SUMS)
    act = (P.slices != -1) & (ps != 0)
    s = P.slices.astype(np.float32)
    r = rng.uniform(-1, 1, P.slices.shape).astype(np.float32)
    simp = np.where(act, s - r, 0.0).astype(np.float32)           # residual e = s*1 - simp (RC.cu:442-447)
    e = np.where(act & (simp > 0), s.astype(np.float64) - simp.astype(np.float64), 0.0)
    rec.debug_set(E.BUF_SIMSLICES, simp)
    rec.debug_set(E.BUF_WEIGHTS, np.ones(P.slices.shape, np.float32))
    rec.SuperresolutionBackproject(np.ones(P.ns, np.float32))
    addon = rec.debug_get(E.BUF_ADDON).astype(np.float64)
    lhs = float(np.sum(sim * sw * e))
    rhs = float(np.sum(addon * V.astype(np.float64)))
    scale = float(np.sum(np.abs(sim * sw * e)))                    # the sums cancel: compare against sum |terms|
    assert abs(lhs - rhs) <= 2e-5 * scale, (lhs, rhs, scale)
    c = rec.counters()
    assert c["Va"] == int(act.sum()) and c["Nv"] == nv


def test_kernel_variants_agree_at_full_size():
    """P4 (too big for the oracle): the production kernels against their simplest variants on the device -- unit-based gather
    with the dead-unit shortcut vs the wave-per-pixel kernel (a unit wrongly declared dead would lose >= 1e-5 of a pixel's
    weight), two-pass Gaussian reconstruction vs the wave-per-pixel kernel, wave-owned / workgroup scatter vs direct atomics.
    Hit sets exact, sums to float round-off."""
    from fetalreconstruction_amd import engine as E, workloads
    P = workloads.get("P4")                                       # the bench's workload: the bundled mask's oblique frame
    rec = _engine(P)
    rec.UpdateScaleVector(np.ones(P.ns), np.ones(P.ns))
    rec.InitializeEMValues()
    out = {}
    rec.set_option("legacy_kernels", 1)
    for name, opts in (("units", dict(gauss_mode=1, fwd_mode=1)), ("simple", dict(gauss_mode=0, fwd_mode=0))):
        for k, v in opts.items():
            rec.set_option(k, v)
        rec.GaussianReconstruction()
        ps = rec.debug_get(E.BUF_PSF_SUMS).copy()
        vol, vw = rec.syncCPU().copy(), rec.getVolWeights().copy()
        rec.SimulateSlices()
        out[name] = (ps, vol, vw, rec.debug_get(E.BUF_SIMSLICES).copy(), rec.debug_get(E.BUF_SIMWEIGHTS).copy(),
                     rec.debug_get(E.BUF_SIMINSIDE).copy(), rec.debug_get(E.BUF_VOXEL_COUNT).copy())
    ref = out["simple"]
    for name in ("units",):
        ps, vol, vw, sim, sw, si, vc = out[name]
        assert np.array_equal(ps != 0, ref[0] != 0) and np.array_equal(vc, ref[6]) and np.array_equal(si, ref[5])
        assert np.allclose(ps, ref[0], rtol=2e-6, atol=0)
        assert rel_err(vol, ref[1]) < TOL_SUM and rel_err(vw, ref[2]) < TOL_SUM     # float atomics in run-dependent order
        assert np.abs(sw - ref[4]).max() < 3e-6 and rel_err(sim, ref[3]) < 5e-6
    rec.set_option("legacy_kernels", 1); rec.set_option("fwd_mode", 1)
    rec.set_option("gauss_mode", 1)
    res = {}
    for bm in (5, 4, 3, 0):
        rec.set_option("legacy_kernels", 1); rec.set_option("back_mode", bm)
        rec.SuperresolutionBackproject(np.ones(P.ns, np.float32))
        res[bm] = (rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy())
    for bm in (5, 4, 3):
        assert np.array_equal(res[bm][1] > 0, res[0][1] > 0)
        assert rel_err(res[bm][1], res[0][1]) < TOL_SUM and rel_err(res[bm][0], res[0][0]) < TOL_SUM


def test_coefficient_table_against_the_oracle(tiny, oracle_mod):
    """Option coeff_table 1: the taps of every live unit are written once (k_coeff_build) and the scatter / gather stream them
    instead of evaluating them -- irtkReconstruction::CoeffInit's _volcoeffs (RG.cc:2305-2673) on the GPU path.  Against the
    oracle like the on-the-fly kernels; the gather bit-identical to the on-the-fly gather (same taps, same order of sums)."""
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    assert rec.get_option("coeff_table") == 1 and rec.get_option("coeff_lazy") == 1     # round 6: the default of a slice-to-volume context
    rec.set_option("coeff_table", 0)                                     # the reference of this test: every tap evaluated
    run_to_state(dg, "sim")
    sim0, sw0 = rec.debug_get(E.BUF_SIMSLICES).copy(), rec.debug_get(E.BUF_SIMWEIGHTS).copy()
    rec.set_option("coeff_table", 1)
    assert rec.get_option("coeff_valid") == 0
    rec.SimulateSlices()                                                 # ... this gather evaluates too, and WRITES the table (coeff_lazy)
    assert rec.get_option("coeff_valid") == 1
    assert np.array_equal(rec.debug_get(E.BUF_SIMSLICES), sim0) and np.array_equal(rec.debug_get(E.BUF_SIMWEIGHTS), sw0)
    rec.SimulateSlices()                                                 # ... and this one reads it
    assert np.array_equal(rec.debug_get(E.BUF_SIMSLICES), sim0) and np.array_equal(rec.debug_get(E.BUF_SIMWEIGHTS), sw0)
    rec.set_option("coeff_lazy", 0); rec.set_option("coeff_invalidate", 1)
    rec.SimulateSlices()                                                 # the table written by k_coeff_build instead: the same bits
    assert rec.get_option("coeff_valid") == 1 and rec.timers()["coeff_build"][1] >= 0
    assert np.array_equal(rec.debug_get(E.BUF_SIMSLICES), sim0) and np.array_equal(rec.debug_get(E.BUF_SIMWEIGHTS), sw0)
    rec.set_option("coeff_lazy", 1)
    run_to_state(do, "sim")
    assert np.array_equal(rec.debug_get(E.BUF_SIMINSIDE), orc.siminside)
    assert rel_err(rec.debug_get(E.BUF_SIMSLICES), orc.simslices) < TOL_SUM
    run_to_state(dg, "scale")
    run_to_state(do, "scale")
    rec.debug_set(E.BUF_WEIGHTS, orc.weights)
    rec.debug_set(E.BUF_SIMSLICES, orc.simslices)
    rec.debug_set(E.BUF_PSF_SUMS, orc.psf_sums)                          # (invalidates the table: it is rebuilt below)
    rec.UpdateScaleVector(orc.d_scales, orc.slice_weights)
    rec.SuperresolutionBackproject(orc.slice_weights)
    orc.SuperresolutionBackproject(orc.slice_weights)
    cm, ad = rec.debug_get(E.BUF_CONFIDENCE_MAP), rec.debug_get(E.BUF_ADDON)
    assert np.array_equal(cm > 0, orc.cmap > 0)
    assert rel_err(cm, orc.cmap) < TOL_SUM and rel_err(ad, orc.addon) < TOL_SUM
    # a whole outer iteration with the table on, each side on its own state
    E, rec, orc, dg, do = _drivers(tiny, oracle_mod)
    rec.set_option("coeff_table", 1)
    dg.reconstruct_iteration(2)
    do.reconstruct_iteration(2)
    g, o = rec.syncCPU(), orc.recon
    assert np.array_equal(g == -1, o == -1) and rel_err(g, o) < 1e-4


def test_coefficient_table_that_does_not_fit_is_switched_off(tiny, monkeypatch):
    """16 KiB per PSF pixel may exceed the free memory (or the ceiling SVR_COEFF_MAX_GB): the engine then evaluates on the fly and
    svr_get_option says so -- same results, no error."""
    from fetalreconstruction_amd import engine as E
    monkeypatch.setenv("SVR_COEFF_MAX_GB", "0.000001")
    rec = _engine(tiny)
    rec.UpdateScaleVector(np.ones(tiny.ns), np.ones(tiny.ns))
    rec.InitializeEMValues()
    rec.GaussianReconstruction()
    rec.SimulateSlices()
    ref = rec.debug_get(E.BUF_SIMSLICES).copy()
    rec.set_option("coeff_table", 1)
    rec.SimulateSlices()
    assert rec.get_option("coeff_table") == 0 and np.array_equal(rec.debug_get(E.BUF_SIMSLICES), ref)


@pytest.mark.parametrize("workload", ["P4", "S8"])
def test_coefficient_table_at_full_size(workload):
    """BASELINE configs[1] / configs[3] on one GPU: the streamed taps give the on-the-fly kernels' results -- the gather bit for
    bit, the scatter up to the order of its float atomics, hit sets exact -- also after new slice matrices (the table follows)."""
    from fetalreconstruction_amd import engine as E, workloads
    P = workloads.get(workload)
    rec = _engine(P)
    rec.UpdateScaleVector(np.ones(P.ns), np.ones(P.ns))
    rec.InitializeEMValues()
    rec.GaussianReconstruction()
    def both():
        rec.SimulateSlices()
        sim, sw, si = (rec.debug_get(b).copy() for b in (E.BUF_SIMSLICES, E.BUF_SIMWEIGHTS, E.BUF_SIMINSIDE))
        rec.SuperresolutionBackproject(np.ones(P.ns, np.float32))
        return sim, sw, si, rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy()
    rec.set_option("coeff_table", 0)
    ref = both()
    rec.set_option("coeff_table", 1)
    both()                                                               # (its gather writes the table, its scatter reads it)
    tab = both()
    assert rec.get_option("coeff_table") != 1 or rec.get_option("coeff_valid") == 1
    if rec.get_option("coeff_table") != 1:                               # P4: 19 GB, S8: 174 GB
        assert workload == "S8"
        pytest.skip("the 174 GB table of S8 does not fit the free memory of this device")
    assert np.array_equal(tab[0], ref[0]) and np.array_equal(tab[1], ref[1]) and np.array_equal(tab[2], ref[2])
    assert np.array_equal(tab[4] > 0, ref[4] > 0)
    assert rel_err(tab[3], ref[3]) < TOL_SUM and rel_err(tab[4], ref[4]) < TOL_SUM
    if workload == "P4":
        t = P.slice_t.copy().reshape(P.ns, 4, 4)
        t[::3, :3, 3] += 0.37                                            # a third of the slices move
        ti = np.stack([np.linalg.inv(m.astype(np.float64)) for m in t]).astype(np.float32)
        rec.SetSliceMatrices(t.reshape(P.ns, 16), ti.reshape(P.ns, 16), P.slice_i2w, P.slice_w2i, P.slice_i2w, P.slice_w2i, P.recon_i2w, P.recon_w2i)
        rec.timer_enable(True); rec.timer_reset()
        rec.GaussianReconstruction()                                     # new matrices: the table follows -- pass 2, which evaluates anyway, writes it
        assert rec.get_option("coeff_valid") == 1
        both()
        tab2 = both()
        rec.set_option("coeff_table", 0)
        ref2 = both()
        assert not np.array_equal(ref2[0], ref[0])
        assert np.array_equal(tab2[0], ref2[0]) and np.array_equal(tab2[1], ref2[1])
        assert np.array_equal(tab2[4] > 0, ref2[4] > 0) and rel_err(tab2[4], ref2[4]) < TOL_SUM


def test_tile_shapes_do_not_change_results(tiny, oracle_mod):
    """The gather's and the scatter's tile shapes are chosen per problem (from the geometry; timed with svr_set_option "fwd_autotune" 1): the simulated slices
    must not depend on the shape at all (fixed per-pixel summation order), the scatter only through the order of its float
    atomics -- and every shape must still agree with the oracle."""
    from fetalreconstruction_amd import engine as E
    orc = oracle_mod.OracleReconstruction(tiny, oracle_mod.CANON)
    ones = np.ones(tiny.ns, np.float32)
    orc.UpdateScaleVector(ones, ones)
    orc.InitializeEMValues()
    orc.GaussianReconstruction()
    orc.SimulateSlices()
    orc.SuperresolutionBackproject(ones)
    res = {}
    for name, opts in (("auto", []), ("8x4/4x4", [("fwd_tile_w", 8), ("fwd_tile_h", 4), ("tile_w", 4), ("tile_h", 4)]),
                       ("4x2/2x2", [("fwd_tile_w", 4), ("fwd_tile_h", 2), ("tile_w", 2), ("tile_h", 2)]),
                       ("2x2/4x2", [("fwd_tile_w", 2), ("fwd_tile_h", 2), ("tile_w", 4), ("tile_h", 2)])):
        rec = _engine(tiny)
        for k, v in opts:
            rec.set_option(k, v)
        rec.UpdateScaleVector(ones, ones)
        rec.InitializeEMValues()
        rec.GaussianReconstruction()
        rec.debug_set(E.BUF_RECONSTRUCTED, orc.recon.astype(np.float32))        # the same volume under every shape
        rec.SimulateSlices()
        sim, sw = rec.debug_get(E.BUF_SIMSLICES).copy(), rec.debug_get(E.BUF_SIMWEIGHTS).copy()
        rec.debug_set(E.BUF_SIMSLICES, orc.simslices)
        rec.SuperresolutionBackproject(ones)
        res[name] = (sim, sw, rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy())
        assert rel_err(sim, orc.simslices) < TOL_SUM and rel_err(res[name][2], orc.addon) < TOL_SUM
        assert np.array_equal(res[name][3] > 0, orc.cmap > 0)
    for name in res:
        assert np.array_equal(res[name][0], res["auto"][0]) and np.array_equal(res[name][1], res["auto"][1]), name
        assert rel_err(res[name][2], res["auto"][2]) < TOL_SUM and rel_err(res[name][3], res["auto"][3]) < TOL_SUM, name


def test_cell_scatter_is_bit_identical_from_run_to_run():
    """back_mode 5 (the default for SVR on the fly; csrc/svr_cell.inc): no float atomics anywhere -- every (cell, plane) box is
    accumulated by one wavefront in a fixed order, staged, and the slabs are added per voxel in a fixed order.  Two launches
    on one engine and a launch on a second engine give the same bits, in the Gaussian pass and in the back-projection (the
    atomic flush of back_mode 4 agrees with itself to ~1e-6 only)."""
    from fetalreconstruction_amd import engine as E, workloads
    P = workloads.get("P4")
    outs = []
    for k in range(2):
        rec = _engine(P)
        assert rec.get_option("back_mode") == 5
        rec.UpdateScaleVector(np.ones(P.ns), np.ones(P.ns))
        rec.InitializeEMValues()
        rec.GaussianReconstruction()
        vol, vw = rec.syncCPU().copy(), rec.getVolWeights().copy()
        rec.SimulateSlices()
        rec.debug_set(E.BUF_WEIGHTS, np.where(P.slices != -1, 0.75, 0).astype(np.float32))   # (the Gaussian pass cleared them, RC.cu:2402)
        for rep in range(2 if k == 0 else 1):
            rec.SuperresolutionBackproject(np.ones(P.ns, np.float32))
            outs.append((vol, vw, rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy()))
        rec.close()
    for n, o in enumerate(outs[1:]):
        for name, a, b in zip(("recon", "volw", "addon", "cmap"), o, outs[0]):
            assert np.array_equal(a, b, equal_nan=True), (n, name, int((a != b).sum()), float(np.nanmax(np.abs(a - b))))
    assert (outs[0][3] > 0).sum() > 100000


@pytest.mark.parametrize("part", ["whole", "a rank's eighth"])
def test_the_three_forms_of_the_combine_give_the_same_bits_at_full_size(part):
    """k_cell_combine (general), k_cell_combine_fast (a voxel's slabs in two batches) and k_cell_combine_wave (round 5, the default: one
    item_of load per wavefront and class, early out where nothing is staged) add the same slabs in the same order: addon | cmap are the
    same bits on the bench workload -- whole, and with a rank's share of the slices (the r-th eighth of every stack, where most of the
    volume has nothing staged around it and the early out is what runs)."""
    from fetalreconstruction_amd import engine as E, phantom, workloads
    from fetalreconstruction_amd.sharding import shard_units
    P = workloads.get("P4")
    if part != "whole":
        order, ranges = shard_units((P.slices != -1).reshape(P.ns, -1).sum(1), P.stack_index, 8, "spatial")
        P = phantom.sub_problem(P, 0, 0, select=order[ranges[3][0]:ranges[3][1]])
    rec = _engine(P)
    rec.UpdateScaleVector(np.ones(P.ns), np.ones(P.ns))
    rec.InitializeEMValues()
    rec.GaussianReconstruction()
    rec.SimulateSlices()
    rec.debug_set(E.BUF_WEIGHTS, np.where(P.slices != -1, 0.75, 0).astype(np.float32))
    outs = {}
    for form in (2, 1, 0):
        rec.set_option("cell_combine", form)
        assert rec.get_option("cell_combine") == form
        rec.SuperresolutionBackproject(np.ones(P.ns, np.float32))
        outs[form] = (rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy())
    rec.close()
    for form in (1, 0):
        for name, a, b in zip(("addon", "cmap"), outs[form], outs[2]):
            assert np.array_equal(a, b, equal_nan=True), (form, name, int((a != b).sum()))
    assert (outs[2][1] > 0).sum() > (100000 if part == "whole" else 10000)


@pytest.mark.parametrize("workload", ["tiny", "P4"])
def test_cell_gather_gives_the_tile_gathers_bits(tiny, workload):
    """fwd_mode 2 against fwd_mode 1: per unit and per pixel the operations and their order are the same (x taps in order, the
    DPP tree over a unit's 16 rows, a pixel's units 0 .. 15), so the simulated slices, weights and inside flags are the same
    bits -- on the tiny problem and on the bench workload."""
    from fetalreconstruction_amd import engine as E, workloads
    P = tiny if workload == "tiny" else workloads.get("P4")
    rec = _engine(P)
    rec.UpdateScaleVector(np.ones(P.ns), np.ones(P.ns))
    rec.InitializeEMValues()
    rec.GaussianReconstruction()
    out = {}
    for mode in (1, 2):
        rec.set_option("legacy_kernels", 1); rec.set_option("fwd_mode", mode)
        for b in (E.BUF_SIMSLICES, E.BUF_SIMWEIGHTS):
            rec.debug_set(b, np.zeros(P.slices.shape, np.float32))
        rec.debug_set(E.BUF_SIMINSIDE, np.zeros(P.slices.shape, np.uint8))
        inside = rec.SimulateSlices()
        out[mode] = (rec.debug_get(E.BUF_SIMSLICES).copy(), rec.debug_get(E.BUF_SIMWEIGHTS).copy(), rec.debug_get(E.BUF_SIMINSIDE).copy(), np.asarray(inside).copy())
    for a, b in zip(out[1], out[2]):
        assert np.array_equal(a, b)
    assert (out[2][1] > 0).sum() > 0.9 * ((P.slices != -1) & (rec.debug_get(E.BUF_PSF_SUMS) != 0)).sum()


@pytest.mark.parametrize("workload", ["tiny", "P4"])
def test_cell_pass_one_gives_the_tile_kernels_bits(tiny, workload):
    """Pass 1 of the Gaussian reconstruction over the (cell, plane) items (fwd_cell_kernel<.., G1>, the default) against the same pass
    per slice tile (fwd_unit_kernel<GAUSS1>, fwd_mode 1): every tap is added in double in the same order (x taps, the tree over a
    unit's rows, a pixel's units 0 .. 15), so v_PSF_sums, the gate, the voxel-count flags -- and with them the reconstructed volume
    and its weights -- are the same bits."""
    from fetalreconstruction_amd import engine as E, workloads
    P = tiny if workload == "tiny" else workloads.get("P4")
    out = {}
    for mode in (1, 2):
        rec = _engine(P)
        rec.set_option("legacy_kernels", 1); rec.set_option("fwd_mode", mode)
        rec.UpdateScaleVector(np.ones(P.ns), np.ones(P.ns))
        rec.InitializeEMValues()
        n = rec.GaussianReconstruction()
        out[mode] = (rec.debug_get(E.BUF_PSF_SUMS).copy(), rec.debug_get(E.BUF_VOXEL_COUNT).copy(), rec.getVolWeights().copy(), rec.syncCPU().copy(), n)
        rec.close()
    for a, b in zip(out[1][:4], out[2][:4]):
        assert np.array_equal(a, b)
    assert out[1][4] == out[2][4] and (out[2][0] != 0).sum() > 0.9 * (P.slices != -1).sum()
