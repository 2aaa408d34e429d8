"""Slice-to-volume NCC cost: the oracle's literal restatement of
irtkImageRigidRegistrationWithPadding::Evaluate against an independent numpy evaluation (CPU),
and the HIP kernel against the oracle (GPU; the six integer moments must be identical)."""
import numpy as np
import pytest

from fetalreconstruction_amd import geometry as geo


def _case(tiny, oracle_mod, n_eval=24, seed=0):
    o = oracle_mod.OracleReconstruction(tiny, oracle_mod.CANON)
    o.GaussianReconstruction()
    vol = o.recon.reshape(tiny.vsize[::-1])
    source = vol.astype(np.int16)                         # static_cast<short>
    targets = np.where(tiny.slices >= 0, tiny.slices, -1).astype(np.int16)
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, tiny.ns, n_eval).astype(np.int32)
    mats = np.zeros((n_eval, 4, 4))
    w2i = tiny.recon_w2i.reshape(4, 4).astype(np.float64)
    for e, k in enumerate(idx):
        t = tiny.slice_t[k].reshape(4, 4).astype(np.float64)
        pert = geo.rigid_matrix(*rng.uniform(-3, 3, 3), *rng.uniform(-4, 4, 3))
        mats[e] = w2i @ (pert @ t) @ tiny.slice_i2w[k].reshape(4, 4).astype(np.float64)
    return source, targets, idx, mats


def _numpy_ncc(target, M, source):
    """Independent evaluation: direct positions M @ (i, j, 0, 1), vectorised trilinear, IRTK round."""
    ty, tx = target.shape
    vz, vy, vx = source.shape
    jj, ii = np.meshgrid(np.arange(ty), np.arange(tx), indexing="ij")
    X = M[0, 0] * ii + M[0, 1] * jj + M[0, 3]
    Y = M[1, 0] * ii + M[1, 1] * jj + M[1, 3]
    Z = M[2, 0] * ii + M[2, 1] * jj + M[2, 3]
    ok = (target >= 0) & (X > 0) & (X < vx - 1) & (Y > 0) & (Y < vy - 1) & (Z > 0) & (Z < vz - 1)
    a, b, c = X[ok].astype(int), Y[ok].astype(int), Z[ok].astype(int)
    t1, u1, v1 = X[ok] - a, Y[ok] - b, Z[ok] - c
    t2, u2, v2 = 1 - t1, 1 - u1, 1 - v1
    s = source.astype(np.float64)
    val = (t1 * (u2 * (v2 * s[c, b, a + 1] + v1 * s[c + 1, b, a + 1]) + u1 * (v2 * s[c, b + 1, a + 1] + v1 * s[c + 1, b + 1, a + 1])) +
           t2 * (u2 * (v2 * s[c, b, a] + v1 * s[c + 1, b, a]) + u1 * (v2 * s[c, b + 1, a] + v1 * s[c + 1, b + 1, a])))
    keep = val >= 0
    sv = np.where(val > 0, (val + 0.5).astype(np.int64), (val - 0.5).astype(np.int64))[keep]
    tv = target[ok][keep].astype(np.int64)
    return np.array([len(tv), tv.sum(), sv.sum(), (tv * tv).sum(), (sv * sv).sum(), (tv * sv).sum()], np.int64)


def test_oracle_ncc_matches_independent_numpy(tiny, oracle_mod):
    source, targets, idx, mats = _case(tiny, oracle_mod)
    nonzero = 0
    for k, M in zip(idx, mats):
        v, sums = oracle_mod.ncc_evaluate(targets[k], M, source)
        ref = _numpy_ncc(targets[k], M, source)
        assert np.array_equal(sums.astype(np.int64), ref)
        if ref[0] > 0:
            nonzero += 1
            n, x, y, x2, y2, xy = ref.astype(np.float64)
            assert v == pytest.approx((xy - x * y / n) / (np.sqrt(x2 - x * x / n) * np.sqrt(y2 - y * y / n)), rel=1e-12)
    assert nonzero >= len(idx) // 2


def test_oracle_ncc_prefers_the_true_alignment(tiny, oracle_mod):
    source, targets, _, _ = _case(tiny, oracle_mod)
    k = int(np.argmax((targets >= 0).reshape(tiny.ns, -1).sum(1)))
    w2i = tiny.recon_w2i.reshape(4, 4).astype(np.float64)
    i2w = tiny.slice_i2w[k].reshape(4, 4).astype(np.float64)
    t = tiny.slice_t[k].reshape(4, 4).astype(np.float64)
    good, _ = oracle_mod.ncc_evaluate(targets[k], w2i @ t @ i2w, source)
    bad, _ = oracle_mod.ncc_evaluate(targets[k], w2i @ (geo.rigid_matrix(tx=4.0, rz=12.0) @ t) @ i2w, source)
    assert good > 0.9 and good > bad + 0.05


@pytest.mark.gpu
def test_device_ncc_moments_equal_the_oracle(tiny, oracle_mod):
    from fetalreconstruction_amd import engine
    source, targets, idx, mats = _case(tiny, oracle_mod, n_eval=64, seed=3)
    rec = engine.Reconstruction(0)
    rec.ncc_set_targets(targets)
    rec.ncc_set_source(source)
    ncc, sums = rec.ncc_evaluate(idx, mats)
    for e, (k, M) in enumerate(zip(idx, mats)):
        v, s = oracle_mod.ncc_evaluate(targets[k], M, source)
        assert np.array_equal(sums[e], s.astype(np.int64))            # integer moments: exact
        assert ncc[e] == pytest.approx(v, rel=1e-12, abs=1e-15)


@pytest.mark.gpu
def test_device_ncc_source_from_reconstruction(tiny, oracle_mod):
    """source = NULL takes static_cast<short> of the engine's current volume (RG.cc:2031)."""
    from fetalreconstruction_amd import engine
    source, targets, idx, mats = _case(tiny, oracle_mod, n_eval=8, seed=5)
    rec = engine.Reconstruction(0)
    engine.sync_gpu(rec, tiny)
    vol = np.random.default_rng(1).uniform(-5, 900, tiny.nvox).astype(np.float32)
    rec.debug_set(engine.BUF_RECONSTRUCTED, vol)
    rec.ncc_set_targets(targets)
    rec.ncc_set_source(None)
    ncc, sums = rec.ncc_evaluate(idx, mats)
    src = vol.reshape(tiny.vsize[::-1]).astype(np.int16)
    for e, (k, M) in enumerate(zip(idx, mats)):
        _, s = oracle_mod.ncc_evaluate(targets[k], M, src)
        assert np.array_equal(sums[e], s.astype(np.int64))
