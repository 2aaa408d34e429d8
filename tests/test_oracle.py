"""CPU tests of the oracle: pinned against the committed golden fixture, its two PSF modes
against each other, and the reference quirks it must reproduce."""
import os

import numpy as np
import pytest

from fetalreconstruction_amd import geometry as geo
from fetalreconstruction_amd import phantom
from fetalreconstruction_amd.sharding import shard_slices
from tests.twins.reconstruction import irtkReconstruction
from tests.util import popcount_xor, rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_v1.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def canon_gauss(tiny, oracle_mod):
    o = oracle_mod.OracleReconstruction(tiny, oracle_mod.CANON)
    o.UpdateScaleVector(np.ones(tiny.ns), np.ones(tiny.ns))
    o.InitializeEMValues()
    n = o.GaussianReconstruction()
    return o, n


def test_problem_is_the_seeded_one(tiny, gold):
    assert float(tiny.slices.astype(np.float64).sum()) == float(gold["slices_sum"])
    assert float(tiny.mask.sum()) == float(gold["mask_sum"])


def test_gaussian_reconstruction_matches_golden(canon_gauss, gold):
    o, n = canon_gauss
    assert np.array_equal(o.psf_sums, gold["psf_sums"])
    assert np.array_equal(o.voxcount.astype(np.uint8), gold["voxcount"])
    assert np.array_equal(o.recon, gold["gauss_recon"])
    assert np.array_equal(o.volw, gold["gauss_volw"])
    assert n[0] == int(gold["voxcount"].sum())


def test_simulate_matches_golden(canon_gauss, gold):
    o, _ = canon_gauss
    inside = o.SimulateSlices()
    assert np.array_equal(o.simslices, gold["simslices0"])
    assert np.array_equal(o.simweights, gold["simweights0"])
    assert np.array_equal(o.siminside, gold["siminside0"])
    assert inside.all()


def test_tap_census_matches_golden(tiny, oracle_mod, gold):
    o = oracle_mod.OracleReconstruction(tiny, oracle_mod.CANON)
    for p, bits in zip(gold["census_pix"], gold["census_bits"]):
        n, b, _, _ = o.tap_census(p[0], p[2], p[1])
        assert np.array_equal(b, bits)
        assert n == popcount_xor(b, np.zeros(64, np.uint64))


def test_literal_and_canonical_psf_agree(tiny, oracle_mod):
    """The canonical float32 sequence (what the HIP kernels run) against the literal
    getPSFParamsPrecomp + calcPSF sequence with libm: values to ~1e-6, skip decisions equal
    up to float-marginal ties."""
    lit = oracle_mod.OracleReconstruction(tiny, oracle_mod.LITERAL)
    can = oracle_mod.OracleReconstruction(tiny, oracle_mod.CANON)
    act = np.argwhere(tiny.slices != -1)
    rng = np.random.default_rng(3)
    kept = flips = 0
    worst = 0.0
    for i in rng.choice(len(act), 120, replace=False):
        sl, py, px = act[i]
        vl, vc = lit.psf_values(sl, px, py), can.psf_values(sl, px, py)
        worst = max(worst, float(np.nanmax(np.abs(vl - vc))))
        nl, bl, _, cl = lit.tap_census(sl, px, py)
        nc, bc, _, cc = can.tap_census(sl, px, py)
        assert np.array_equal(cl, cc)           # centre voxel: index work, identical
        kept += nl
        flips += popcount_xor(bl, bc)
    assert worst < 5e-6                         # PSF values are O(1)
    assert flips <= 2e-4 * kept                 # epsilon-skip ties


def test_flip_pixels_is_the_two_census_walks_compared(tiny, oracle_mod):
    """orc_flip_pixels (what the whole-workload LITERAL comparisons attribute their outliers to, round 6): per pixel the number of taps the literal walk
    processes and the canonical one skips or the other way round -- the xor of the two tap_census keep masks -- and the PSF mass behind them; on the
    pixels with a flip, v_PSF_sums of the two modes differ by no more than that mass (+ the two sequences' value differences)."""
    lit = oracle_mod.OracleReconstruction(tiny, oracle_mod.LITERAL)
    can = oracle_mod.OracleReconstruction(tiny, oracle_mod.CANON)
    act = np.argwhere(tiny.slices != -1)
    pick = act[np.random.default_rng(5).choice(len(act), 400, replace=False)]
    flips, open_, mass = can.flip_pixels(pick)
    f2, _, m2 = lit.flip_pixels(pick)
    assert np.array_equal(flips, f2) and np.array_equal(mass, m2)          # (both sequences are walked whatever the instance's own mode)
    for (sl, py, px), f, m in zip(pick, flips, mass):
        _, bl, vl, _ = lit.tap_census(sl, px, py, with_vals=True)
        _, bc, vc, _ = can.tap_census(sl, px, py, with_vals=True)
        assert popcount_xor(bl, bc) == f
        if f:
            kl = np.unpackbits(bl.view(np.uint8), bitorder="little").astype(bool)
            kc = np.unpackbits(bc.view(np.uint8), bitorder="little").astype(bool)
            x = kl != kc
            assert abs(float(np.maximum(np.where(kl, vl, 0), np.where(kc, vc, 0))[x].sum()) - m) <= 1e-5 * max(m, 1e-6) + 1e-7
            assert abs(float(vl[kl].sum()) - float(vc[kc].sum())) <= 1.05 * m + 3e-3 * max(float(vl[kl].sum()), 1.0)
    assert (open_ >= 0).all() and flips.sum() <= 2e-4 * 4096 * len(pick)


def test_literal_and_canonical_pipeline_agree(tiny, oracle_mod):
    outs = []
    for mode in (oracle_mod.LITERAL, oracle_mod.CANON):
        o = oracle_mod.OracleReconstruction(tiny, mode)
        o.UpdateScaleVector(np.ones(tiny.ns), np.ones(tiny.ns))
        o.InitializeEMValues()
        o.GaussianReconstruction()
        o.SimulateSlices()
        outs.append(o)
    a, b = outs
    assert np.array_equal(a.psf_sums > 0, b.psf_sums > 0)
    assert rel_err(a.psf_sums, b.psf_sums) < 2e-3       # a flipped tap moves sume by one psf value
    assert rel_err(a.recon, b.recon) < 1e-3
    assert rel_err(a.simslices, b.simslices) < 1e-3


def test_negative_coordinates_alias_to_index_zero(oracle_mod):
    """float->uint saturation (RC.cu:241,274,382,508): a slice hanging off the low corner of the
    volume still deposits into x==0 / y==0 / z==0 voxels."""
    P = phantom.make_problem(1, (12, 12, 2), 1.0, 2.0, None, 1.0, 14.0, seed=5, orientations=("ax",),
                             motion_frac=0.0, noise_sigma=0.0)
    # push the stack towards the low corner: taps with negative coordinates appear
    shift = geo.rigid_matrix(tx=-14.2, ty=-14.4, tz=-14.6)
    for k in range(P.ns):
        P.slice_t[k] = geo.to_matrix4(shift)
        P.slice_tinv[k] = geo.to_matrix4(np.linalg.inv(shift))
    P.mask[...] = 1.0
    P.slices[...] = 100.0
    o = oracle_mod.OracleReconstruction(P, oracle_mod.CANON)
    o.GaussianReconstruction()
    vol = o.volw.reshape(P.vsize[::-1])
    assert (o.psf_sums > 0).any()
    # the faces at index 0 collect every aliased tap: far more weight than the next plane
    assert vol[0].sum() > 3 * vol[1].sum() or vol[:, 0].sum() > 3 * vol[:, 1].sum() or vol[:, :, 0].sum() > 3 * vol[:, :, 1].sum()


def test_voxel_aligned_pixels_are_dropped_by_nan_psf(oracle_mod):
    """sin(R)/R is NaN at R == 0 (RC.cu:129): an exactly voxel-aligned slice loses every pixel
    (sume is NaN, RC.cu:251-258)."""
    P = phantom.make_problem(1, (10, 10, 1), 1.0, 2.0, None, 1.0, 12.0, seed=5, orientations=("ax",),
                             motion_frac=0.0, noise_sigma=0.0, stack_offsets_mm=0.0)
    ident = np.eye(4)
    for k in range(P.ns):
        P.slice_t[k] = geo.to_matrix4(ident)
        P.slice_tinv[k] = geo.to_matrix4(ident)
    P.slices[...] = 50.0
    v = np.array(P.vsize)
    assert ((v[:2] - 1) % 2 == (np.array([10, 10]) - 1) % 2).all()   # grids share in-plane voxel centres
    for mode in (oracle_mod.LITERAL, oracle_mod.CANON):
        o = oracle_mod.OracleReconstruction(P, mode)
        n = o.GaussianReconstruction()
        assert n == [0] and not (o.psf_sums != 0).any()


def test_psf_sums_persist_across_gaussian_passes(tiny, oracle_mod):
    """v_PSF_sums is never cleared (RC.cu:2401-2411): a stale value survives a later pass in which
    the pixel's sume drops to <= 0.5."""
    o = oracle_mod.OracleReconstruction(tiny, oracle_mod.CANON)
    o.GaussianReconstruction()
    before = o.psf_sums.copy()
    far = geo.rigid_matrix(tx=500.0)
    o._keep[2][0] = geo.to_matrix4(far)                    # slice 0 leaves the volume entirely
    o._keep[3][0] = geo.to_matrix4(np.linalg.inv(far))
    o.GaussianReconstruction()
    assert np.array_equal(o.psf_sums[0], before[0]) and (before[0] > 0).any()


def test_scale_vector_lags_one_call_on_the_device(tiny, oracle_mod):
    o = oracle_mod.OracleReconstruction(tiny, oracle_mod.CANON)
    r = irtkReconstruction(o, tiny.ns, max_intensity=tiny.max_intensity, min_intensity=tiny.min_intensity)
    from tests.util import run_to_state
    run_to_state(r, "estep0")
    r.ScaleGPU()
    assert np.array_equal(o.d_scales, np.ones(tiny.ns, np.float32))      # still the initial vector
    assert np.array_equal(o.h_scales, r._scale_gpu) and (r._scale_gpu != 1).any()


def test_host_estep_matches_c_restatement(tiny, oracle_mod):
    rng = np.random.default_rng(11)
    for trial in range(20):
        ns = int(rng.integers(3, 40))
        pot = rng.uniform(0, 0.6, ns).astype(np.float32)
        pot[rng.random(ns) < 0.15] = -1
        w = rng.uniform(0, 1, ns).astype(np.float32) if trial % 3 else np.ones(ns, np.float32)
        sc = rng.uniform(0.1, 6.0 if trial % 4 == 0 else 1.5, ns).astype(np.float32)

        class Fake:
            def EStep(self, m, s, x):
                return pot.copy()

            def UpdateSliceWeights(self, w_):
                self.w = np.array(w_)

        r = irtkReconstruction(Fake(), ns)
        r._slice_weight_gpu = w.copy()
        r._scale_gpu = sc.copy()
        r._mix_s_gpu = 0.9 if trial % 2 else 0.7
        st = np.array([0, 0, r._sigma_s_gpu, r._sigma_s2_gpu, r._mix_s_gpu], np.float32)
        pot_c, w_c, st_c = oracle_mod.host_estep(pot, w, sc, [], [], 0.0001, st)
        r.EStepGPU()
        assert np.array_equal(r._slice_potential_gpu, pot_c)
        assert np.allclose(r._slice_weight_gpu, w_c, rtol=1e-6, atol=1e-7)
        assert np.allclose([r._mean_s_gpu, r._mean_s2_gpu, r._sigma_s_gpu, r._sigma_s2_gpu, r._mix_s_gpu], st_c,
                           rtol=1e-6, atol=1e-9)


def test_shard_slices_is_a_balanced_partition():
    rng = np.random.default_rng(2)
    for world in (1, 2, 3, 8):
        a = rng.integers(0, 5000, 97)
        r = shard_slices(a, world)
        assert r[0][0] == 0 and r[-1][1] == 97
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        loads = [a[lo:hi].sum() for lo, hi in r]
        assert max(loads) <= a.sum() / world + a.max()


def test_slice_cost_weights_follow_the_live_planes():
    """Sharding weights: a slice whose normal is the volume's x axis keeps all 16 planes of its footprints (x is the axis of the
    sequential skip chain and cannot be owned), an axial one about 12; thicker slices keep more."""
    from fetalreconstruction_amd.sharding import slice_cost_weights
    def i2w(normal_axis):
        m = np.eye(4)
        cols = {2: [0, 1, 2], 0: [1, 2, 0], 1: [2, 0, 1]}[normal_axis]      # slice x, y, z directions in world axes
        r = np.zeros((3, 3))
        for j, c in enumerate(cols):
            r[c, j] = 1.0
        m[:3, :3] = r
        return m.reshape(16)
    eye = np.eye(4).reshape(16)
    w = slice_cost_weights([100, 100, 100, 100], [i2w(2), i2w(1), i2w(0), i2w(2)], [eye] * 4, eye,
                           [[1, 1, 2.5], [1, 1, 2.5], [1, 1, 2.5], [1, 1, 5.0]], 1.0)
    assert np.isclose(w[0], w[1]) and np.isclose(w[2], 100 * (9.4 + 16.0) * 1.2)
    assert np.isclose(w[0], 100 * (9.4 + 2 * 5.1 * 2.5 / 2.3548 + 1)) and w[3] > w[0]


def test_geometry_conventions():
    a = geo.ImageAttributes(10, 12, 5, 1.2, 0.9, 2.5, np.array([0, 1.0, 0]), np.array([0, 0, 1.0]),
                            np.array([1.0, 0, 0]), np.array([3.0, -2.0, 7.5]))
    i2w, w2i = geo.image_to_world(a), geo.world_to_image(a)
    assert np.allclose(i2w @ w2i, np.eye(4), atol=1e-12)
    c = i2w @ np.array([4.5, 5.5, 2.0, 1.0])
    assert np.allclose(c[:3], a.origin)                     # the origin is the image centre
    t = geo.rigid_matrix(1, 2, 3, 10, 20, 30)
    assert np.allclose(t[:3, :3] @ t[:3, :3].T, np.eye(3), atol=1e-12)
    assert np.allclose(geo.psf_centre_offset((1.0, 1.0, 1.0)), 0.0)
    assert np.all(np.abs(geo.psf_centre_offset((0.8, 0.8, 0.8))) < 1e-5)
