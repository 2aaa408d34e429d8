"""SLICO superpixels and the superpixel patches of the patch-based path (fetalreconstruction_amd/slic.py)."""
import numpy as np
import pytest

from fetalreconstruction_amd import geometry as geo
from fetalreconstruction_amd import phantom
from tests.twins import pvr, slic


def _stacks():
    return phantom.make_stacks(2, (48, 40, 6), 1.1, 2.2, None, 1.0, 16.0, seed=4, orientations=("ax", "sag"), stack_motion_mm=0.0,
                               stack_motion_deg=0.0)


def test_lab_conversion_and_seeds():
    l, a, b = slic.rgbtolab(np.array([0, 128, 255]), np.array([0, 128, 255]), np.array([0, 128, 255]))
    assert np.allclose(l, [0.0, 53.585, 100.0], atol=1e-2) and np.allclose(a, 0, atol=5e-3) and np.allclose(b, 0, atol=5e-3)   # greys
    s = slic.get_seeds(12, 40, 48)                                  # 3 x 4 strips of a 40 x 48 image, spread over the remainder
    assert len(s) == 12 and s[0] == 6 * 40 + 6 and (np.diff(s) > 0).all()
    assert set(s % 40) == {6, 19, 32} and set(s // 40) == {6, 18, 30, 42}


def test_slico_labels_are_connected_and_cover_the_slice():
    from scipy import ndimage
    stacks, *_ = _stacks()
    data = np.asarray(stacks[0].data, np.float32)
    lab = slic.segment_slic(data, (12, 12))
    assert lab.shape == data.shape and lab.min() == 0
    for z in range(data.shape[0]):
        n = int(lab[z].max()) + 1
        assert 8 <= n <= 16                                         # 48 * 40 / 144 = 13 requested
        for k in range(n):
            comp, cnt = ndimage.label(lab[z] == k)
            assert cnt == 1, (z, k)                                 # EnforceSuperpixelConnectivity
        assert min(np.bincount(lab[z].astype(int).ravel())) > 144 // 4
    assert np.array_equal(lab, slic.segment_slic(data, (12, 12)))   # deterministic
    # the reference runs SLIC on the transposed slice (x outer, y inner buffer): a transposed input gives the transposed labels
    # of a run whose seed grid is laid out for the other orientation -- not the same partition as the untransposed run
    one = slic.segment_slic(data[:1].transpose(0, 2, 1).copy(), (12, 12))[0].T
    assert one.shape == lab[0].shape


def test_superpixel_patches_and_masks():
    stacks, mask, mattr, rattr, rmask = _stacks()
    st = stacks[0]
    p, i2w, w2i, m, org, attrs = slic.generate2DSuperpixelPatches(st, mask, mattr, (12, 12), 30)
    p0, _, _, m0, _, _ = slic.generate2DSuperpixelPatches(st, mask, mattr, (12, 12), 0)
    assert p.shape[1:] == (40, 48) and len(p) == len(p0) > 20        # 64 x 64 clamped to the slice; dilation does not change the count
    lab = slic.segment_slic(np.asarray(st.data, np.float32), (12, 12))
    assert len(p) <= sum(int(lab[z].max()) for z in range(lab.shape[0]))         # the largest label of a slice is never cut out
    mm, mm0 = m.reshape(-1, 64, 64), m0.reshape(-1, 64, 64)
    for k in range(len(p)):
        on = mm[k][:40, :48] == ord("1")
        assert np.array_equal(on, p[k] != -1) and not mm[k][40:].any() and not mm[k][:, 48:].any()   # spxMask[i + 64 j] = '1' where the patch has a value
        assert on.sum() >= (mm0[k] == ord("1")).sum() >= 144 // 4                                     # dilated by 30 % of the superpixel's larger side
    # a patch pixel with a value carries the stack pixel under it
    s_w2i = geo.world_to_image(st.attr)
    for k in (0, len(p) // 2, len(p) - 1):
        ys, xs = np.nonzero(p[k] != -1)
        for j, i in list(zip(ys, xs))[::37]:
            q = s_w2i @ (i2w[k].reshape(4, 4).astype(np.float64) @ np.array([i, j, 0, 1.0]))
            x, y, z = [int(round(v)) for v in q[:3]]
            assert p[k][j, i] == np.float32(st.data[z, y, x])
        assert np.allclose(w2i[k].reshape(4, 4) @ i2w[k].reshape(4, 4), np.eye(4), atol=1e-4)
    P = pvr.make_pvr_problem(stacks, mask, mattr, rattr, rmask, (12, 12), (30, 30), superpixel=True)
    assert P.spx_masks.shape == (P.ns, 4096) and P.slices.shape[1:] == (40, 48) and len(P.patch_ri2w) == P.ns == len(P.slice_attr)


def test_pvr_command_line_with_superpixels_on_the_oracle(tmp_path, oracle_mod):
    from fetalreconstruction_amd import nifti
    from tests.twins import pvr_cli
    stacks, mask, mattr, rattr, rmask = _stacks()
    paths = []
    for k, st in enumerate(stacks):
        nifti.write(tmp_path / f"s{k}.nii.gz", st.data, st.attr)
        paths.append(str(tmp_path / f"s{k}.nii.gz"))
    nifti.write(tmp_path / "mask.nii.gz", rmask, rattr)
    seen = {}

    def factory(prob, device):
        seen["prob"] = prob
        return oracle_mod.OracleReconstruction(prob, oracle_mod.CANON, pvr=True, spx_masks=prob.spx_masks)

    out = tmp_path / "o.nii.gz"
    assert pvr_cli.main(["-o", str(out), "-i", *paths, "-m", str(tmp_path / "mask.nii.gz"), "--superpixel", "--spxSize", "12", "--spxExtend", "30",
                         "--resolution", "1.0", "--iterations", "0", "--sr_iterations", "1", "--no_registration"], _engine_factory=factory) == 0
    P = seen["prob"]
    assert P.spx_masks is not None and P.ns > 30
    vol, va = nifti.read(out)
    kk, jj, ii = np.meshgrid(np.arange(va.nz), np.arange(va.ny), np.arange(va.nx), indexing="ij")
    w = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(float) @ geo.image_to_world(va).T
    inside = (np.sum(w[..., :3] ** 2, -1) < 12.0 ** 2) & (vol > 0)
    assert inside.sum() > 3000
    assert np.corrcoef(vol[inside], phantom.phantom_intensity(w[..., :3], 16.0)[inside])[0, 1] > 0.6


@pytest.mark.gpu
def test_pvr_command_line_with_superpixels_end_to_end(tmp_path):
    from fetalreconstruction_amd import nifti
    from tests.twins import pvr_cli
    stacks, mask, mattr, rattr, rmask = _stacks()
    paths = []
    for k, st in enumerate(stacks):
        nifti.write(tmp_path / f"s{k}.nii.gz", st.data, st.attr)
        paths.append(str(tmp_path / f"s{k}.nii.gz"))
    nifti.write(tmp_path / "mask.nii.gz", rmask, rattr)
    out = tmp_path / "o.nii.gz"
    assert pvr_cli.main(["-o", str(out), "-i", *paths, "-m", str(tmp_path / "mask.nii.gz"), "-s", "--spxSize", "12", "--spxExtend", "30",
                         "--resolution", "1.0", "--iterations", "1", "--sr_iterations", "3"]) == 0
    vol, va = nifti.read(out)
    kk, jj, ii = np.meshgrid(np.arange(va.nz), np.arange(va.ny), np.arange(va.nx), indexing="ij")
    w = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(float) @ geo.image_to_world(va).T
    inside = (np.sum(w[..., :3] ** 2, -1) < 12.0 ** 2) & (vol > 0)
    assert inside.sum() > 3000 and np.corrcoef(vol[inside], phantom.phantom_intensity(w[..., :3], 16.0)[inside])[0, 1] > 0.5   # 6-slice stacks of 4.4 mm patches: coarse
    # the C++ command line (csrc/svr_slic.h) gives the same volume
    import subprocess
    from fetalreconstruction_amd import build
    r = subprocess.run([build.PVR_CLI, "-o", str(tmp_path / "cc.nii.gz"), "-i", *paths, "-m", str(tmp_path / "mask.nii.gz"), "-s", "--spxSize", "12",
                        "--spxExtend", "30", "--resolution", "1.0", "--iterations", "1", "--sr_iterations", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    vc, _ = nifti.read(tmp_path / "cc.nii.gz")
    ok = (vol > 0) & (vc > 0)
    assert vc.shape == vol.shape and np.corrcoef(vol[ok], vc[ok])[0, 1] > 0.97


def test_cpp_superpixel_patches_match_the_python_ones(tmp_path):
    """bin/PVRreconstructionGPU -s --dumpProblem --dryRun (csrc/svr_slic.h) against slic.py through the same pre-processing."""
    import subprocess
    from fetalreconstruction_amd import build, nifti
    from tests.twins import pvr_cli
    from fetalreconstruction_amd import preprocess as pp
    build.build()
    stacks, mask, mattr, rattr, rmask = _stacks()
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
    # the Python twin on the same files
    ims = []
    for p in paths:
        d, at = nifti.read(p)
        ims.append(pp.Image(d.astype(np.float64), at))
    md, mat = nifti.read(tmp_path / "mask.nii.gz")
    ims, ts, iso, tattr, rm = pvr_cli.prepare(ims, [np.eye(4)] * 2, pp.Image(md.astype(np.float64), mat), 1.0, 0, False)
    pst = [pvr.Stack(s.data.astype(np.float32), s.attr, t, s.attr.dz) for s, t in zip(ims, ts)]
    P = pvr.make_pvr_problem(pst, iso.data, iso.attr, tattr, rm.data, (12, 12), (30, 30), superpixel=True)
    assert list(counts) == list(P.patches_per_stack) and (py, px) == P.slices.shape[1:]
    assert np.array_equal(masks, P.spx_masks) and np.array_equal(patches, P.slices)     # same labels, same dilation, same values
