"""The host functions either side of the hot path against an independent restatement of the reference's loops
(oracle/prep_oracle.c) -- not only against their twin:

  MatchStackIntensitiesWithMasking (RG.cc:1375-1493): preprocess.py (what cli.py runs) and, through the command line's problem
      dump, csrc/svr_prep.h,
  generate2DPatches (patchBasedObject.cuh:174-342): pvr.py and csrc/pvr_cli.cpp,
  segmentSLIC (runStackSLIC.cpp:55-840) + generate2DSuperpixelPatches (patchBasedObject.cuh:347-367, 433-802): slic.py and
      csrc/svr_slic.h,

on axis-aligned stacks and on the reference's bundled mask geometry (oblique, 300-400 mm off the origin)."""
import subprocess

import numpy as np
import pytest

import real_mask as rm
from fetalreconstruction_amd import geometry as geo
from fetalreconstruction_amd import phantom
from tests.twins import pvr
from fetalreconstruction_amd import preprocess as pp


def _oblique_case(n=3):
    m, a, _ = rm.load()
    stacks, c = rm.stacks_on_mask_grid(m, a, n)
    rng = np.random.default_rng(3)
    ts = [geo.rigid_matrix(*np.concatenate([rng.uniform(-2, 2, 3), rng.uniform(-2, 2, 3)])) if k else np.eye(4) for k in range(n)]
    return m, a, stacks, ts


def _aligned_case():
    stacks, mask, mattr, rattr, rmask = phantom.make_stacks(3, (32, 32, 8), 1.1, 2.2, None, 1.0, 14.0, seed=3)
    return stacks, mask, mattr


@pytest.mark.parametrize("together", [False, True])
def test_match_stack_intensities_against_the_oracle(oracle_mod, together):
    m, a, stacks, ts = _oblique_case()
    imgs = [pp.Image(d.astype(np.float64), sa) for d, sa in stacks]
    factors = pp.MatchStackIntensitiesWithMasking(imgs, ts, pp.Image(m.astype(np.float64), a), 700.0, together=together)
    data, fac, avg = oracle_mod.match_stack_intensities([d for d, _ in stacks], [sa for _, sa in stacks], ts, m, a, 700.0, together)
    # the averages are double sums in different orders (numpy pairwise, the reference x-outermost): 1e-12; the factors are
    # stored as float like _stack_factor
    assert np.allclose(factors.astype(np.float64), fac.astype(np.float64), rtol=2e-7, atol=0)
    for img, d in zip(imgs, data):
        assert np.array_equal(img.data > 0, d > 0)
        assert np.allclose(img.data, d, rtol=1e-12, atol=0)
    if together:
        assert len(set(np.round(fac.astype(np.float64) / fac[0], 6))) == 1
    else:
        inside = [float(d[(d > 0)].mean()) for d in data]
        assert np.all(np.array(inside) > 0)
    # the rule itself: after matching, the mean over the ROI is the requested value
    data1, fac1, avg1 = oracle_mod.match_stack_intensities(data, [sa for _, sa in stacks], ts, m, a, 700.0, together)
    assert np.allclose(avg1 if not together else np.mean(avg1), 700.0, rtol=1e-9)


def test_stack_without_overlap_is_an_error(oracle_mod):
    m, a, stacks, ts = _oblique_case(2)
    far = geo.rigid_matrix(tx=1000.0)
    with pytest.raises(ValueError):
        oracle_mod.match_stack_intensities([d for d, _ in stacks], [sa for _, sa in stacks], [np.eye(4), far], m, a, 700.0)
    with pytest.raises(ValueError):
        pp.MatchStackIntensitiesWithMasking([pp.Image(d.astype(np.float64), sa) for d, sa in stacks], [np.eye(4), far],
                                            pp.Image(m.astype(np.float64), a), 700.0)


@pytest.mark.parametrize("case", ["aligned", "oblique", "full_slices"])
def test_generate_2d_patches_against_the_oracle(oracle_mod, case):
    if case == "oblique":
        m, a, st, _ = _oblique_case(2)
        stacks = [pvr.Stack(d.astype(np.float32), sa, np.eye(4), sa.dz) for d, sa in st]
        mask, mattr = (m > 0).astype(np.uint8), a
        size, stride, full = (16, 16), (8, 8), False
    else:
        stacks, mask, mattr = _aligned_case()
        size, stride, full = ((16, 16), (8, 8), False) if case == "aligned" else ((0, 0), (0, 0), True)
    checked = 0
    for st in stacks:
        if full:
            pbb = (st.attr.nx, st.attr.ny)
            p, i2w, w2i, total, org = pvr.generate2DPatches(st, mask, mattr, pbb, (pbb[0] + 1, pbb[1] + 1), with_origins=True)
        else:
            p, i2w, w2i, total, org = pvr.generate2DPatches(st, mask, mattr, size, stride, with_origins=True)
        op, oi, ow, ototal, oorg = oracle_mod.generate_2d_patches(st.data, st.attr, st.thickness, mask, mattr, size, stride, full_slices=full, snap=True)
        assert len(op) == len(p) > 0 and ototal == total
        assert np.array_equal(op, p)                                   # which pixel every patch pixel reads: exact
        assert np.allclose(oi, i2w, rtol=0, atol=2e-4) and np.allclose(ow, w2i, rtol=0, atol=2e-4)    # float32 of doubles built in two orders, 300 mm off
        assert np.allclose(oorg, org, rtol=0, atol=1e-9)
        checked += len(p)
        if case == "aligned":
            # on a grid whose arithmetic is exact the literal truncation and the 1e-6 snap agree
            lp, *_ = oracle_mod.generate_2d_patches(st.data, st.attr, st.thickness, mask, mattr, size, stride, snap=False)
            assert len(lp) == len(op)
    assert checked > 10


def test_cpp_patches_against_the_oracle(tmp_path, oracle_mod):
    """bin/PVRreconstructionGPU --dumpProblem --dryRun: the patches the C++ command line cuts (csrc/pvr_cli.cpp) are the oracle's
    cut of the same pre-processed stacks."""
    from fetalreconstruction_amd import build, nifti
    from tests.test_pvr import _python_pvr_problem, _write_pvr_case
    build.build()
    paths, mpath, stacks = _write_pvr_case(tmp_path)
    dump = tmp_path / "problem.bin"
    r = subprocess.run([build.PVR_CLI, "-o", str(tmp_path / "x.nii.gz"), "-i", *paths, "-m", mpath, "--patchSize", "16", "16", "--patchStride", "8", "8",
                        "--resolution", "1.0", "--no_registration", "--dumpProblem", str(dump), "--dryRun"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    raw = dump.read_bytes()
    hdr = np.frombuffer(raw, np.int32, 8)
    ns, px, py, nst = [int(v) for v in hdr[:4]]
    o = 32
    counts = np.frombuffer(raw, np.int32, nst, o); o += 4 * nst + 8
    patches = np.frombuffer(raw, np.float32, ns * py * px, o).reshape(ns, py, px); o += 4 * ns * py * px
    i2w = np.frombuffer(raw, np.float32, ns * 16, o).reshape(ns, 16)
    # the same pre-processed stacks (mask binarisation, cropping, iso mask, intensity matching: the Python chain, which the
    # dump test of test_pvr.py ties to the C++ one), cut by the oracle
    P, _, _ = _python_pvr_problem(paths, mpath, (16, 16), (8, 8), 1.0)
    from tests.twins import pvr_cli
    md, mat = nifti.read(mpath)
    ims = [pp.Image(nifti.read(p)[0].astype(np.float64), nifti.read(p)[1]) for p in paths]
    st2, ts, iso_mask, tattr, recon_mask = pvr_cli.prepare(ims, [np.eye(4)] * len(ims), pp.Image(md.astype(np.float64), mat), 1.0, 0, False)
    at = 0
    for k, s in enumerate(st2):
        op, oi, ow, _, _ = oracle_mod.generate_2d_patches(s.data.astype(np.float32), s.attr, s.attr.dz, iso_mask.data, iso_mask.attr, (16, 16), (8, 8))
        assert len(op) == counts[k]
        assert np.array_equal(op, patches[at:at + len(op)])
        assert np.allclose(oi, i2w[at:at + len(op)], atol=1e-5)
        at += len(op)
    assert at == ns


@pytest.mark.parametrize("case", ["aligned", "oblique"])
def test_slico_labels_and_superpixel_patches_against_the_oracle(oracle_mod, case):
    """SLICO labels (seeds, ten zero-parameter iterations on the transposed slice, connectivity pass) and the 64 x 64 patches cut
    around the superpixels with their dilated spxMask: slic.py against the oracle's literal loops."""
    from tests.twins import slic
    if case == "oblique":
        m, a, st, _ = _oblique_case(2)
        stacks = [pvr.Stack(d.astype(np.float32), sa, np.eye(4), sa.dz) for d, sa in st]
        mask, mattr = (m > 0).astype(np.uint8), a
        spx, ext = (16, 16), 20
    else:
        stacks, mask, mattr, _, _ = phantom.make_stacks(2, (48, 40, 6), 1.1, 2.2, None, 1.0, 16.0, seed=4, orientations=("ax", "sag"),
                                                        stack_motion_mm=0.0, stack_motion_deg=0.0)
        spx, ext = (12, 12), 30
    checked = 0
    for st in stacks:
        data = np.asarray(st.data, np.float32)
        lab = slic.segment_slic(data, spx)
        olab = oracle_mod.segment_slic(data, spx)
        assert np.array_equal(lab, olab)                                 # every pixel's superpixel: exact
        p, i2w, w2i, msk, org, _ = slic.generate2DSuperpixelPatches(st, mask, mattr, spx, ext)
        op, om, oi, ow, oorg, total = oracle_mod.generate_2d_superpixel_patches(data, st.attr, olab, st.thickness, mask, mattr, spx, ext)
        assert len(op) == len(p) > 0
        assert np.array_equal(op, p) and np.array_equal(om, msk)         # which pixels a patch carries, their values, the 64-wide mask
        assert total == int((msk == ord("1")).sum())
        assert np.allclose(oi, i2w, rtol=0, atol=2e-4) and np.allclose(ow, w2i, rtol=0, atol=2e-4)
        assert np.allclose(oorg, org, rtol=0, atol=1e-9)
        checked += len(p)
    assert checked > 20


def test_cpp_superpixel_patches_against_the_oracle(tmp_path, oracle_mod):
    """bin/PVRreconstructionGPU -s --dumpProblem --dryRun: the superpixel patches and masks the C++ command line cuts
    (csrc/svr_slic.h) are the oracle's cut of the same pre-processed stacks."""
    from fetalreconstruction_amd import build, nifti
    from tests.twins import pvr_cli
    build.build()
    stacks, mask, mattr, rattr, rmask = phantom.make_stacks(2, (48, 40, 6), 1.1, 2.2, None, 1.0, 16.0, seed=4, orientations=("ax", "sag"),
                                                            stack_motion_mm=0.0, stack_motion_deg=0.0)
    paths = []
    for k, st in enumerate(stacks):
        nifti.write(tmp_path / f"s{k}.nii.gz", st.data, st.attr)
        paths.append(str(tmp_path / f"s{k}.nii.gz"))
    nifti.write(tmp_path / "mask.nii.gz", rmask, rattr)
    dump = tmp_path / "problem.bin"
    r = subprocess.run([build.PVR_CLI, "-o", str(tmp_path / "x.nii.gz"), "-i", *paths, "-m", str(tmp_path / "mask.nii.gz"), "-s", "--spxSize", "12",
                        "--spxExtend", "30", "--resolution", "1.0", "--no_registration", "--dumpProblem", str(dump), "--dryRun"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    raw = dump.read_bytes()
    ns, px, py, nst, vx, vy, vz = [int(v) for v in np.frombuffer(raw, np.int32, 8)[:7]]
    o = 32
    counts = np.frombuffer(raw, np.int32, nst, o); o += 4 * nst + 8
    patches = np.frombuffer(raw, np.float32, ns * py * px, o).reshape(ns, py, px); o += 4 * ns * py * px + 64 * ns + 4 * vx * vy * vz
    masks = np.frombuffer(raw, np.uint8, ns * 4096, o).reshape(ns, 4096)
    # the same pre-processed stacks (the Python chain, tied to the C++ one by the dump tests), cut by the oracle
    ims = [pp.Image(nifti.read(p)[0].astype(np.float64), nifti.read(p)[1]) for p in paths]
    md, mat = nifti.read(tmp_path / "mask.nii.gz")
    ims, ts, iso, tattr, rmk = pvr_cli.prepare(ims, [np.eye(4)] * 2, pp.Image(md.astype(np.float64), mat), 1.0, 0, False)
    at = 0
    for k, s in enumerate(ims):
        data = s.data.astype(np.float32)
        olab = oracle_mod.segment_slic(data, (12, 12))
        op, om, *_ = oracle_mod.generate_2d_superpixel_patches(data, s.attr, olab, s.attr.dz, iso.data, iso.attr, (12, 12), 30)
        assert len(op) == counts[k]
        assert np.array_equal(op, patches[at:at + len(op)]) and np.array_equal(om, masks[at:at + len(op)])
        at += len(op)
    assert at == ns > 20


# ---- CreateTemplate / SetMask / TransformMask / CropImage / MaskSlices: the C++ command line against the oracle's restatement ----
def _read_svr_dump(path):
    import ctypes as C
    from oracle import pyoracle as po
    raw = open(path, "rb").read()
    hdr = np.frombuffer(raw, np.int32, 8)
    ns, mx, my, n, tx, ty, tz, ver = [int(v) for v in hdr]
    assert ver == 2
    o = 32
    asz = C.sizeof(po.Attr)
    tattr = po._attr_to_py(po.Attr.from_buffer_copy(raw[o:o + asz])); o += asz
    vmask = np.frombuffer(raw, np.float64, tx * ty * tz, o).reshape(tz, ty, tx); o += 8 * tx * ty * tz
    sattrs = []
    for _ in range(n):
        sattrs.append(po._attr_to_py(po.Attr.from_buffer_copy(raw[o:o + asz]))); o += asz
    grid = np.frombuffer(raw, np.float32, ns * my * mx, o).reshape(ns, my, mx); o += 4 * ns * my * mx
    T = np.frombuffer(raw, np.float64, 16 * ns, o).reshape(ns, 4, 4); o += 128 * ns
    fac = np.frombuffer(raw, np.float32, n, o); o += 4 * n
    sx = np.frombuffer(raw, np.int32, ns, o); o += 4 * ns
    sy = np.frombuffer(raw, np.int32, ns, o); o += 4 * ns
    assert o == len(raw)
    return dict(tattr=tattr, vmask=vmask, sattrs=sattrs, grid=grid, T=T, factors=fac, sizes_x=sx, sizes_y=sy)


def _same_attr(a, b, tol=1e-9):
    return ((a.nx, a.ny, a.nz) == (b.nx, b.ny, b.nz) and np.allclose([a.dx, a.dy, a.dz], [b.dx, b.dy, b.dz], rtol=1e-12)
            and np.allclose(a.origin, b.origin, atol=tol) and np.allclose(np.stack([a.xaxis, a.yaxis, a.zaxis]), np.stack([b.xaxis, b.yaxis, b.zaxis]), atol=1e-12))


@pytest.mark.parametrize("case,smooth", [("oblique", 4.0), ("oblique", 0.0), ("aligned", 2.0)])
def test_cpp_preprocessing_chain_against_the_oracle(tmp_path, oracle_mod, case, smooth):
    """bin/SVRreconstructionGPU --dumpProblem --dryRun (csrc/svr_prep.h, csrc/svr_cli.cpp) against the oracle's restatement of the
    reference's loops, step by step: TransformMask + CropImage of the template stack (RG.cc:805-821, 5205-5306), CreateTemplate
    (RG.cc:648-694), SetMask with and without smoothing (RG.cc:750-803), TransformMask + CropImage of the other stacks,
    MatchStackIntensitiesWithMasking (RG.cc:1375-1493), MaskSlices (RG.cc:1940-1988) -- on the reference's bundled mask geometry
    (oblique, 300-400 mm off the origin, where an x.5 coordinate decides a voxel) and on an axis-aligned case."""
    from fetalreconstruction_amd import build, nifti
    po = oracle_mod
    build.build()
    if case == "oblique":
        m, a, st, _ = _oblique_case(3)
        stacks = [(d.astype(np.float64), sa) for d, sa in st]
        mask, mattr = m.astype(np.float64), a
        res, thick = 1.0, 2.5
    else:
        sts, mask_, mattr = _aligned_case()
        stacks = [(s.data.astype(np.float64), s.attr) for s in sts]
        mask = np.asarray(mask_, np.float64)
        res, thick = 1.0, 2.2
    paths = []
    for k, (d, sa) in enumerate(stacks):
        nifti.write(tmp_path / f"s{k}.nii", d.astype(np.float32), sa)
        paths.append(str(tmp_path / f"s{k}.nii"))
    nifti.write(tmp_path / "mask.nii", mask.astype(np.float32), mattr)
    dump = tmp_path / "svr.bin"
    r = subprocess.run([build.CLI, "-o", str(tmp_path / "x.nii"), "-i", *paths, "-m", str(tmp_path / "mask.nii"), "--thickness", *[str(thick)] * len(paths),
                        "--resolution", str(res), "--smooth_mask", str(smooth), "--no_registration", "--dumpProblem", str(dump), "--dryRun"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    D = _read_svr_dump(str(dump))
    # start the oracle chain from what the files hold: float32 voxels, and the geometry as NIfTI stores it
    stacks = []
    for pth in paths:
        d_, a_ = nifti.read(pth)
        stacks.append((d_.astype(np.float64), a_))
    mask_f, mattr = nifti.read(tmp_path / "mask.nii")
    mask = mask_f.astype(np.float64)
    I = np.eye(4)
    # 1. the template stack: TransformMask (with the UNSMOOTHED mask file) + CropImage
    d0, a0 = stacks[0]
    m0 = po.transform_mask(d0, a0, mask, mattr, I)
    d0c, a0c, _ = po.crop_image(d0, a0, m0)
    assert _same_attr(a0c, D["sattrs"][0])
    # 2. CreateTemplate
    tattr, d = po.create_template(a0c, res)
    assert d == res and _same_attr(tattr, D["tattr"])
    # 3. SetMask: blur + threshold + nearest-neighbour resampling onto the template
    vmask, blurred = po.set_mask(tattr, mask, mattr, smooth)
    assert vmask.shape == D["vmask"].shape and 0 < vmask.sum() < vmask.size
    assert np.array_equal(vmask, D["vmask"])
    if smooth > 0:
        assert not np.array_equal(blurred, mask)                          # the smoothing changed the mask
    # 4. the other stacks: the VOLUME mask transformed onto them, crop
    cropped = [(d0c, a0c)]
    for k in range(1, len(stacks)):
        dk, ak = stacks[k]
        mk = po.transform_mask(dk, ak, vmask, tattr, I)
        dc, ac, _ = po.crop_image(dk, ak, mk)
        assert _same_attr(ac, D["sattrs"][k]), k
        cropped.append((dc, ac))
    # 5. MatchStackIntensitiesWithMasking (already restated in round 2), 6. slices + MaskSlices
    data, fac, avg = po.match_stack_intensities([c[0] for c in cropped], [c[1] for c in cropped], [I] * len(cropped), vmask, tattr, 700.0)
    assert np.allclose(fac, D["factors"], rtol=2e-7)
    import copy
    sl = 0
    for k, ((dc, ac), dm) in enumerate(zip(cropped, data)):
        for j in range(ac.nz):
            sa = po.get_region_attr(ac, 0, 0, j, ac.nx, ac.ny, j + 1)      # CreateSlicesAndTransformations RG.cc:1835-1880
            sa.dz = thick
            ms = po.mask_slice(dm[j], sa, I, vmask, tattr)
            g = D["grid"][sl][:ac.ny, :ac.nx]
            assert D["sizes_x"][sl] == ac.nx and D["sizes_y"][sl] == ac.ny
            assert np.array_equal(g == -1, ms == -1), (k, j, int(((g == -1) != (ms == -1)).sum()))
            assert np.allclose(g, ms.astype(np.float32), rtol=1e-6, atol=0)
            assert (D["grid"][sl][ac.ny:, :] == -1).all() and (D["grid"][sl][:, ac.nx:] == -1).all()   # padding of the grid
            sl += 1
    assert sl == D["grid"].shape[0] and (D["grid"] != -1).sum() > 1000
