"""The reference's default (CPU / IRTK) registration schedule around the batched NCC cost (csrc/irtk_reg.cpp; SURVEY 8a16,
8f1 'the IRTK schedule', 8f2 'stack-to-stack registration').  CPU tests run the C++ schedule over the oracle's restatement
of irtkImageRigidRegistrationWithPadding::Evaluate; the GPU test shows that the engine's evaluator gives the identical
optimisation trajectory (the six moments are exact integers)."""
import numpy as np
import pytest

from fetalreconstruction_amd import geometry as geo
from fetalreconstruction_amd import host, phantom
from tests.twins import registration as reg


def _oracle_backend(oracle_mod, log=None):
    def fn(target, M, source):
        v, sums = oracle_mod.ncc_evaluate(target, M, source)
        if log is not None:
            log.append((target.shape, M.copy(), sums.copy()))
        return sums
    return host.NccBackend(fn)


def _points(radius=9.0, n=200, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.normal(size=(n, 3))
    p = p / np.linalg.norm(p, axis=1, keepdims=True) * rng.uniform(0, radius, (n, 1))
    return np.concatenate([p, np.ones((n, 1))], 1)


def _max_error_mm(a, b, radius=9.0):
    p = _points(radius)
    return float(np.linalg.norm((p @ np.asarray(a).T - p @ np.asarray(b).T)[:, :3], axis=1).max())


# ---- building blocks -----------------------------------------------------------------------------------------------------
def test_rigid_parameters_round_trip():
    rng = np.random.default_rng(0)
    for _ in range(20):
        p = np.concatenate([rng.uniform(-30, 30, 3), rng.uniform(-80, 80, 3)])
        m = geo.rigid_matrix(*p)
        q, rebuilt = host.irtk_rigid_parameters(m)
        assert np.allclose(q, p, atol=1e-9) and np.allclose(rebuilt, m, atol=1e-12)
    # gimbal lock (ry = 90): Matrix2Parameters' second branch, rz = 0 (irtkRigidTransformation.cc:137-142)
    q, rebuilt = host.irtk_rigid_parameters(geo.rigid_matrix(1, 2, 3, 20, 90, 0))
    assert q[5] == 0 and np.allclose(rebuilt, geo.rigid_matrix(1, 2, 3, 20, 90, 0), atol=1e-9)


def test_resampling_with_padding_matches_the_python_mirror():
    rng = np.random.default_rng(1)
    a = geo.ImageAttributes(12, 10, 6, 1.1, 1.1, 2.2, origin=np.array([2.0, -1.0, 0.5]))
    d = rng.integers(1, 900, (6, 10, 12)).astype(np.int16)
    d[:, :2] = -1
    d[0] = -1
    out, oa = host.irtk_resample_with_padding(d, a, (2.2, 2.2, 2.2), -1)
    ref, ra = reg.resample_with_padding(d.astype(np.float64), a, (2.2, 2.2, 2.2), pad=-1.0)
    assert (oa.nx, oa.ny, oa.nz) == (ra.nx, ra.ny, ra.nz) == (6, 5, 6) and oa.dx == 2.2
    assert np.array_equal(out, np.trunc(ref).astype(np.int16))          # PutAsDouble: static_cast, no rounding
    assert (out == -1).any() and (out > 0).any()


def test_blurring_with_padding_rules():
    a = geo.ImageAttributes(15, 9, 1, 1.0, 2.0, 3.0)
    d = np.full((1, 9, 15), 100, np.int16)
    d[0, :, :3] = -5                                                     # <= padding: stays padding, never contributes
    d[0, 4, 8] = 1000
    out = host.irtk_blur_with_padding(d, a, 1.0, -5)
    assert (out[0, :, :3] == -5).all()
    # normalised over the valid taps only: a flat image stays flat -- up to PutAsDouble's truncation after each pass
    # (99.99999.. -> 99, then 98.99.. -> 98)
    assert np.isin(out[0, :2, 3:], (98, 99, 100)).all() and np.isin(out[0, :, 13:], (98, 99, 100)).all()
    assert 100 < out[0, 4, 9] < out[0, 4, 8] < 1000 and out[0, 4, 8 + 5] in (98, 99, 100)   # kernel of 2*round(4 sigma/d)+1 = 9 taps in x
    # independent evaluation of one pixel: x pass (sigma 1 px) then y pass (sigma 0.5 px), each truncated to short
    k = lambda s, n: np.exp(-np.arange(-n, n + 1) ** 2 / (2.0 * s * s))   # noqa: E731
    kx = k(1.0, 4)
    row = lambda y: int((kx * d[0, y, 3:12]).sum() / kx.sum())           # noqa: E731  pixel x = 7 of row y after the x pass
    ky = k(0.5, 1)                                                       # 2 * round(4 * 1 / 2) + 1 = 5 taps -> half 2; outer taps ~ 3e-4
    ky = k(0.5, 2)
    col = np.array([row(y) for y in range(2, 7)], np.float64)
    assert out[0, 4, 7] == int((ky * col).sum() / ky.sum())
    assert a.nz == 1                                                      # single plane: no z pass (GBWP.cc:90)


# ---- stack-to-stack registration ------------------------------------------------------------------------------------------
def _stacks(motion_mm=1.5, motion_deg=2.0, seed=5):
    stacks, mask, mattr, rattr, rmask = phantom.make_stacks(2, (36, 36, 14), 1.1, 2.2, None, 1.0, 13.0, seed=seed, noise_sigma=2.0,
                                                            orientations=("ax", "ax"), stack_motion_mm=motion_mm, stack_motion_deg=motion_deg)
    return stacks, rattr, rmask


def test_stack_registration_recovers_the_stack_motion(oracle_mod):
    stacks, rattr, rmask = _stacks()
    truth = np.linalg.inv(stacks[0].transformation) @ stacks[1].transformation      # stack 1 -> template space
    start = [np.eye(4), np.eye(4)]
    assert _max_error_mm(start[1], truth) > 1.5
    log = []
    be = _oracle_backend(oracle_mod, log)
    t, nev = host.StackRegistrations(None, [s.data.astype(np.float64) for s in stacks], [s.attr for s in stacks], start, 0, backend=be)
    assert np.array_equal(t[0], np.eye(4))                                            # the template is not registered
    err = _max_error_mm(t[1], truth)
    print("stack registration: error before", _max_error_mm(start[1], truth), "after", err, "evaluations", nev)
    assert err < 0.8                                                                  # in-plane voxels of 1.1 mm, 2.2 mm slices; last step 0.25 mm / deg
    assert 100 < nev < 3000 and be.calls < nev                                        # one backend call per optimiser step, not per evaluation
    # the 3-D target goes to the evaluator as planes whose start positions are the iterator's accumulated ones: the moments of
    # the planes of one evaluation add up to the oracle's own 3-D evaluation (irtkHomogeneousTransformationIterator NextZ)
    shapes = [s for s, _, _ in log]
    nz = 14                                                                           # finest level: the template's 14 planes
    k = len(log) - nz
    assert all(s == shapes[k] for s in shapes[k:])
    planes = np.stack([be.targets[i] for i in range(nz)])
    whole, sums3d = oracle_mod.ncc_evaluate(planes, log[k][1], be.source)
    assert np.array_equal(sums3d, sum(s for _, _, s in log[k:]))


def test_stack_registration_uses_the_mask_and_keeps_good_alignment(oracle_mod):
    stacks, rattr, rmask = _stacks(motion_mm=0.0, motion_deg=0.0)
    be = _oracle_backend(oracle_mod)
    t, nev = host.StackRegistrations(None, [s.data.astype(np.float64) for s in stacks], [s.attr for s in stacks], [np.eye(4)] * 2, 0,
                                     mask=rmask.astype(np.float64), mask_attr=rattr, backend=be)
    assert _max_error_mm(t[1], np.eye(4)) < 0.3
    # StackRegistrations zeroes the template outside the mask (RG.cc:959-984) and 0 is the target padding: the finest-level
    # planes are -1 outside the ROI
    assert (be.targets[:, 0, 0] == -1).all() and (be.targets >= 0).any()


# ---- slice-to-volume registration (the reference's default) ------------------------------------------------------------------
def _slice_case(tiny, oracle_mod):
    o = oracle_mod.OracleReconstruction(tiny, oracle_mod.CANON)
    o.InitializeEMValues()
    o.GaussianReconstruction()
    vol = o.recon.reshape(tiny.vsize[::-1]).astype(np.float32)
    vol = np.where(tiny.mask.reshape(vol.shape) > 0, vol, -1).astype(np.float32)       # maskVolume
    rattr = geo.ImageAttributes(*tiny.vsize, *tiny.vdim)
    sel = [4, 9, 12, 20]
    T = np.stack([tiny.slice_t[k].reshape(4, 4).astype(np.float64) for k in sel])
    rng = np.random.default_rng(2)
    P = T.copy()
    for k in (0, 2):
        P[k] = geo.rigid_matrix(*rng.uniform(-1.5, 1.5, 3), *rng.uniform(-2.5, 2.5, 3)) @ T[k]
    return vol, rattr, sel, T, P


def test_slice_to_volume_registration_pulls_slices_back(tiny, oracle_mod):
    vol, rattr, sel, T, P = _slice_case(tiny, oracle_mod)
    be = _oracle_backend(oracle_mod)
    out, nev = host.SliceToVolumeRegistration(None, tiny.slices[sel], [tiny.slice_attr[k] for k in sel], P, rattr, vol, backend=be)
    before = [_max_error_mm(P[k], T[k], 12.0) for k in range(4)]
    after = [_max_error_mm(out[k], T[k], 12.0) for k in range(4)]
    print("slice registration errors before", np.round(before, 2), "after", np.round(after, 2), "evaluations", nev, "calls", be.calls)
    assert after[0] < 0.7 * before[0] and after[2] < 0.7 * before[2]                  # the knocked slices come back
    assert after[1] < 1.0 and after[3] < 1.0                                          # the aligned ones stay
    assert be.calls < nev / 3                                                         # lock step: 4 slices share every call
    # deterministic
    out2, nev2 = host.SliceToVolumeRegistration(None, tiny.slices[sel], [tiny.slice_attr[k] for k in sel], P, rattr, vol,
                                                backend=_oracle_backend(oracle_mod))
    assert np.array_equal(out, out2) and nev == nev2
    # a slice without any valid pixel is left alone (smax > -1 test, RG.cc:2018)
    empty = np.full_like(tiny.slices[sel[:1]], -1)
    out3, nev3 = host.SliceToVolumeRegistration(None, empty, [tiny.slice_attr[sel[0]]], P[:1], rattr, vol, backend=_oracle_backend(oracle_mod))
    assert np.array_equal(out3[0], P[0]) and nev3 == 0


def test_registration_refuses_missing_engine():
    with pytest.raises(Exception):
        host.StackRegistrations(None, [np.zeros((2, 2, 2))], [geo.ImageAttributes(2, 2, 2, 1, 1, 1)], [np.eye(4)], 0)


def _moved_attr(a, G):
    """the same image seen from the frame G (a rigid map of the world)"""
    import copy
    r = copy.copy(a)
    r.xaxis, r.yaxis, r.zaxis = G[:3, :3] @ np.asarray(a.xaxis, float), G[:3, :3] @ np.asarray(a.yaxis, float), G[:3, :3] @ np.asarray(a.zaxis, float)
    r.origin = (G @ np.array([*np.asarray(a.origin, float), 1.0]))[:3]
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("frame", ["phantom", "bundled mask"])
def test_engine_and_oracle_evaluators_give_the_same_trajectory(tiny, oracle_mod, frame):
    """`bundled mask`: the same case in the oblique frame of the reference's bundled mask, 475 mm from the world origin
    (tests/real_mask.py): the decisions of the optimiser still coincide between the device and the oracle similarity."""
    from fetalreconstruction_amd import engine as E
    vol, rattr, sel, T, P = _slice_case(tiny, oracle_mod)
    rec = E.Reconstruction(0)
    sattr = [tiny.slice_attr[k] for k in sel]
    if frame != "phantom":
        import real_mask as rm
        m, a, _ = rm.load()
        G = np.eye(4)
        G[:3, 0], G[:3, 1], G[:3, 2] = a.xaxis, a.yaxis, a.zaxis
        G[:3, 3] = rm.centre(m, a)
        Gi = np.linalg.inv(G)
        sattr = [_moved_attr(x, G) for x in sattr]
        rattr = _moved_attr(rattr, G)
        P = np.stack([G @ p @ Gi for p in P])
    args = (tiny.slices[sel], sattr, P, rattr, vol)
    dev, nev_d = host.SliceToVolumeRegistration(rec, *args)
    cpu, nev_c = host.SliceToVolumeRegistration(None, *args, backend=_oracle_backend(oracle_mod))
    assert nev_d == nev_c and np.array_equal(dev, cpu)                                # exact integer moments -> identical decisions
    assert nev_d > 100 and not np.array_equal(dev, P)
    if frame != "phantom":
        return
    stacks, _, _ = _stacks()
    sargs = ([s.data.astype(np.float64) for s in stacks], [s.attr for s in stacks], [np.eye(4)] * 2, 0)
    dev, nev_d = host.StackRegistrations(rec, *sargs)
    cpu, nev_c = host.StackRegistrations(None, *sargs, backend=_oracle_backend(oracle_mod))
    assert nev_d == nev_c and np.array_equal(dev, cpu)


@pytest.mark.gpu
def test_device_pyramid_is_the_host_pyramid(tiny, oracle_mod, monkeypatch):
    """The engine blurs, resamples and re-ranges the three levels of every target and of the source on the device
    (csrc/svr_pyr.inc); SVR_HOST_PYRAMID=1 makes them with the host code of csrc/irtk_reg.cpp and uploads them.  Same
    integer moments, so the same decisions and the same matrices, for 2-D targets (slices) and 3-D targets (stacks)."""
    from fetalreconstruction_amd import engine as E
    vol, rattr, sel, T, P = _slice_case(tiny, oracle_mod)
    rec = E.Reconstruction(0)
    args = (tiny.slices[sel], [tiny.slice_attr[k] for k in sel], P, rattr, vol)
    stacks, _, _ = _stacks()
    sargs = ([s.data.astype(np.float64) for s in stacks], [s.attr for s in stacks], [np.eye(4)] * 2, 0)
    dev, nev_d = host.SliceToVolumeRegistration(rec, *args)
    sdev, snev_d = host.StackRegistrations(rec, *sargs)
    monkeypatch.setenv("SVR_HOST_PYRAMID", "1")
    cpu, nev_c = host.SliceToVolumeRegistration(rec, *args)
    scpu, snev_c = host.StackRegistrations(rec, *sargs)
    assert nev_d == nev_c and nev_d > 100 and np.array_equal(dev, cpu)
    assert snev_d == snev_c and np.array_equal(sdev, scpu)


# ---- packages (interleaved sub-stacks): PackageToVolume ------------------------------------------------------------------------
def _package_case():
    """one axial stack acquired as 2 interleaved packages that moved differently, and the analytic volume"""
    R = 13.0
    a = geo.ImageAttributes(36, 36, 14, 1.1, 1.1, 2.2)
    t_pack = [geo.rigid_matrix(1.2, -0.8, 0.5, 1.5, -2.0, 1.0), geo.rigid_matrix(-1.0, 0.9, -0.6, -1.0, 1.5, -2.0)]
    kk, jj, ii = np.meshgrid(np.arange(a.nz), np.arange(a.ny), np.arange(a.nx), indexing="ij")
    pix = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(np.float64)
    data = np.zeros((a.nz, a.ny, a.nx))
    for k in range(a.nz):
        w = (pix[k] @ geo.image_to_world(a).T) @ t_pack[k % 2].T
        data[k] = phantom.phantom_intensity(w[..., :3], R) * 700 / 0.55
    ra = geo.ImageAttributes(34, 34, 34, 1.0, 1.0, 1.0)
    kk, jj, ii = np.meshgrid(np.arange(34), np.arange(34), np.arange(34), indexing="ij")
    w = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(np.float64) @ geo.image_to_world(ra).T
    vol = (phantom.phantom_intensity(w[..., :3], R) * 700 / 0.55).astype(np.float32)
    return a, data, t_pack, ra, vol


def test_package_to_volume_recovers_package_motion(oracle_mod):
    a, data, t_pack, ra, vol = _package_case()
    start = np.tile(np.eye(4), (a.nz, 1, 1))
    be = _oracle_backend(oracle_mod)
    t, nev = host.PackageToVolume(None, [data], [a], [2], start, ra, vol, backend=be)
    err = [_max_error_mm(t[k], t_pack[k % 2], 10.0) for k in range(a.nz)]
    before = [_max_error_mm(np.eye(4), t_pack[k % 2], 10.0) for k in range(a.nz)]
    print("package registration: error before", np.round(before[:2], 2), "after", np.round(err[:2], 2), "evaluations", nev, "calls", be.calls)
    assert max(err) < 0.5 * min(before)
    for k in range(2, a.nz):                                    # every slice of a package carries its package's transformation
        assert np.allclose(t[k], t[k % 2], atol=1e-12)
    assert not np.allclose(t[0], t[1], atol=1e-3)
    assert be.calls < nev                                        # the two packages advance in lock step
    # even/odd splitting: 2 x 2 sub-packages, slices 0,4,8,.. / 2,6,.. / 1,5,.. / 3,7,..
    t2, _ = host.PackageToVolume(None, [data], [a], [2], start, ra, vol, evenodd=True, backend=_oracle_backend(oracle_mod))
    groups = {tuple(np.round(m.reshape(-1), 9)) for m in t2}
    assert len(groups) == 4 and np.allclose(t2[0], t2[4]) and np.allclose(t2[1], t2[5]) and not np.allclose(t2[0], t2[2])
    # halves of the even/odd packages (HalfImage: packages of >= 4 slices are cut in two)
    t3, _ = host.PackageToVolume(None, [data], [a], [1], start, ra, vol, evenodd=True, half=True, half_iter=1, backend=_oracle_backend(oracle_mod))
    assert len({tuple(np.round(m.reshape(-1), 9)) for m in t3}) == 4          # 1 package -> even/odd (7 + 7 slices) -> halves (3 + 4 each)


@pytest.mark.parametrize("nz,packages,evenodd,half,half_iter", [(14, 2, False, False, 1), (15, 4, False, False, 1), (14, 2, True, False, 1), (17, 3, True, False, 1),
                                                                 (14, 1, True, True, 1), (29, 2, True, True, 1), (40, 2, True, True, 2), (7, 3, True, True, 1)])
def test_package_splitting_against_the_oracle(oracle_mod, nz, packages, evenodd, half, half_iter):
    """SplitImage / SplitImageEvenOdd / SplitImageEvenOddHalf / HalfImage and the slice assignment of PackageToVolume
    (irtkReconstructionGPU.cc:4980-5192) restated in oracle/prep_oracle.c (orc_split_packages), against the C++ product
    (csrc/irtk_reg.cpp svrh_package_to_volume) on an oblique stack.  The product is observed from outside: every slice starts with
    its own transformation, the similarity backend is flat (no registration step is accepted), so on return every slice of a package
    carries the start transformation of the package's FIRST slice -- the membership of every package and its first slice, which is
    what the oracle lists.  The oracle's two routes to a package's slices must agree too: by construction (plane k of package l is
    plane k * packages + l, halves keep their planes) and by the reference's geometry (ImageToWorld of the package, WorldToImage of
    the stack, round)."""
    rot = geo.rigid_matrix(0, 0, 0, 17.0, -23.0, 31.0)[:3, :3]
    a = geo.ImageAttributes(12, 10, nz, 1.1, 1.3, 2.2, rot[:, 0].copy(), rot[:, 1].copy(), rot[:, 2].copy(), origin=np.array([13.7, -41.2, 88.9]))
    packs = oracle_mod.split_packages(a, packages, evenodd, half, half_iter)
    seen = np.zeros(nz, int)
    for attr, assigned, held in packs:
        assert np.array_equal(assigned, held)                       # the geometry finds the planes the splitting put there
        seen[held] += 1
    assert (seen == 1).all()                                        # every slice in exactly one package
    rng = np.random.default_rng(5)
    data = rng.uniform(100, 900, (nz, a.ny, a.nx))
    start = np.stack([geo.rigid_matrix(0.01 * (k + 1), -0.02 * (k + 1), 0.005 * k, 0.1 * k, 0, 0) for k in range(nz)])
    ra = geo.ImageAttributes(20, 20, 20, 1.0, 1.0, 1.0, origin=a.origin.copy())
    vol = rng.uniform(100, 900, (20, 20, 20)).astype(np.float32)

    # a similarity that never improves: every optimiser stops where it started
    be = host.NccBackend(lambda t, M, s: np.array([1.0, 1.0, 1.0, 1.0, 1.0, 4.0]))
    t, nev = host.PackageToVolume(None, [data], [a], [packages], start, ra, vol, evenodd=evenodd, half=half, half_iter=half_iter, backend=be)
    for attr, assigned, held in packs:
        first = int(assigned[0])
        for sl in assigned:                                         # (the other slices get the first one's PARAMETERS: a rebuilt matrix)
            assert np.allclose(t[sl], start[first], atol=1e-9), (first, sl)
    # distinct packages kept distinct transformations (the start transformations are all different)
    firsts = sorted(int(p[1][0]) for p in packs)
    assert len({tuple(np.round(t[f].reshape(-1), 8)) for f in firsts}) == len(packs)
