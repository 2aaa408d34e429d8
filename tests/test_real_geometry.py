"""The reference bundles one data file, the brain mask of its README example (data/mask_10_3T_brain_smooth.nii.gz): an oblique
288x288x84 acquisition whose box sits 300-400 mm from the world origin.  tests/golden/bundled_mask_bbox.npz keeps its geometry
and the mask voxels of the bounding box (tests/golden/make_mask_fixture.py); these tests run the pre-processing chains and the
command lines on that geometry with synthetic stacks (the 3T stacks themselves are not bundled)."""
import subprocess

import numpy as np
import pytest

import real_mask as rm
from fetalreconstruction_amd import geometry as geo


def test_fixture_is_the_bundled_mask_geometry():
    m, a, f = rm.load()
    assert tuple(f["full_shape"]) == (84, 288, 288) and int(f["count"]) == 318377 == int(m.sum())
    assert np.allclose(f["voxel"], [1.17647, 1.17647, 1.25], atol=1e-5)
    nzv = np.argwhere(m > 0)
    assert tuple(nzv.max(0) - nzv.min(0) + 1) == (70, 93, 100)                  # the 100 x 93 x 70 box of config R
    R = np.stack([a.xaxis, a.yaxis, a.zaxis])
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and np.abs(R).max() < 0.95  # orthonormal, oblique to every world axis
    assert np.linalg.norm(rm.centre(m, a)) > 300.0                                # far from the world origin


def _write_case(tmp_path, n=3):
    from fetalreconstruction_amd import nifti
    m, a, _ = rm.load()
    stacks, c = rm.stacks_on_mask_grid(m, a, n)
    paths = []
    for k, (d, sa) in enumerate(stacks):
        nifti.write(tmp_path / f"s{k}.nii.gz", d, sa)
        paths.append(str(tmp_path / f"s{k}.nii.gz"))
    nifti.write(tmp_path / "mask.nii.gz", m, a)
    return paths, str(tmp_path / "mask.nii.gz"), stacks, c, (m, a)


def test_nifti_round_trip_keeps_the_oblique_geometry(tmp_path):
    from fetalreconstruction_amd import nifti
    paths, mpath, stacks, c, (m, a) = _write_case(tmp_path, 2)
    d, ra = nifti.read(mpath)
    assert np.array_equal(d, m) and np.allclose(geo.image_to_world(ra), geo.image_to_world(a), atol=1e-4)
    for p, (sd, sa) in zip(paths, stacks):
        d, ra = nifti.read(p)
        assert np.array_equal(d, sd) and np.allclose(geo.image_to_world(ra), geo.image_to_world(sa), atol=1e-4)


def _read_dump(dump):
    raw = dump.read_bytes()
    hdr = np.frombuffer(raw, np.int32, 8)
    ns, px, py, nst, vx, vy, vz = [int(v) for v in hdr[:7]]
    o = 32
    counts = np.frombuffer(raw, np.int32, nst, o); o += 4 * nst
    o += 8
    patches = np.frombuffer(raw, np.float32, ns * py * px, o).reshape(ns, py, px); o += 4 * ns * py * px
    i2w = np.frombuffer(raw, np.float32, ns * 16, o).reshape(ns, 16); o += 64 * ns
    mask = np.frombuffer(raw, np.float32, vx * vy * vz, o)
    return counts, patches, i2w, mask, (vx, vy, vz)


def test_patches_on_the_oblique_grid(tmp_path):
    """Patch extraction where no coordinate is exact: a patch pixel sits on a slice pixel, and the double arithmetic of an oblique
    grid puts it 1e-14 to either side (the reference truncates that, patchBasedObject.cuh:262-277).  Both command lines snap such
    coordinates, cut the same patches bit for bit, and every patch pixel is the slice pixel under it."""
    import test_pvr as TP
    from fetalreconstruction_amd import build, nifti
    build.build()
    paths, mpath, stacks, c, _ = _write_case(tmp_path, 3)
    dump = tmp_path / "p.bin"
    r = subprocess.run([build.PVR_CLI, "-o", str(tmp_path / "x.nii.gz"), "-i", *paths, "-m", mpath, "--patchSize", "32", "32", "--patchStride",
                        "16", "16", "--resolution", "1.0", "--no_registration", "--dumpProblem", str(dump), "--dryRun"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    counts, patches, i2w, mask, vsize = _read_dump(dump)
    P, pmin, pmax = TP._python_pvr_problem(paths, mpath, (32, 32), (16, 16), 1.0)
    assert vsize == P.vsize and list(counts) == list(P.patches_per_stack) and min(counts) > 500
    assert np.array_equal(mask, P.mask.reshape(-1)) and np.array_equal(patches, P.slices)
    assert np.allclose(i2w, P.slice_i2w, atol=1e-4)
    # every patch pixel that was copied is the pixel of the (cropped, intensity-matched) stack it sits on
    q = 0
    for cnt, st in zip(counts, P.cropped_stacks):
        w2i = geo.world_to_image(st.attr)
        for k in range(q, q + cnt, 37):
            first = w2i @ (P.slice_i2w[k].reshape(4, 4).astype(np.float64) @ np.array([0, 0, 0, 1.0]))
            x, y, z = (int(v) for v in np.rint(first[:3]))
            assert np.abs(first[:3] - [x, y, z]).max() < 1e-3                 # float32 matrices of a far-away grid
            ys, xs = np.nonzero(patches[k] > 0)
            assert len(ys) > 32 * 32 / 3.0
            assert np.array_equal(patches[k][ys, xs], st.data[z][ys + y, xs + x].astype(np.float32))
        q += cnt


def _correlation(path, c, mask, mattr):
    from fetalreconstruction_amd import nifti
    vol, va = nifti.read(path)
    t = rm.truth(va, c)
    kk, jj, ii = np.meshgrid(np.arange(va.nz), np.arange(va.ny), np.arange(va.nx), indexing="ij")
    w = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(np.float64) @ (geo.world_to_image(mattr) @ geo.image_to_world(va)).T
    m = np.rint(w[..., :3]).astype(int)
    ok = (m >= 0).all(-1) & (m[..., 0] < mattr.nx) & (m[..., 1] < mattr.ny) & (m[..., 2] < mattr.nz)
    inside = np.zeros(vol.shape, bool)
    inside[ok] = mask[m[ok][:, 2], m[ok][:, 1], m[ok][:, 0]] > 0
    inside &= vol > 0
    return vol, va, inside, float(np.corrcoef(vol[inside], t[inside])[0, 1])


@pytest.mark.gpu
def test_svr_command_lines_on_the_oblique_grid(tmp_path):
    """bin/SVRreconstructionGPU and cli.py on three mutually oblique stacks on the bundled mask's grid, 300-400 mm from the
    origin: the reconstruction lands on the mask (float32 transform chains hold up) and shows the phantom."""
    from fetalreconstruction_amd import build
    from tests.twins import cli
    paths, mpath, stacks, c, (m, a) = _write_case(tmp_path, 3)
    common = ["-i", *paths, "-m", mpath, "--thickness", "2.5", "2.5", "2.5", "--resolution", "1.0", "--iterations", "2",
              "--rec_iterations_first", "3", "--rec_iterations_last", "5", "--no_registration"]
    r = subprocess.run([build.CLI, "-o", str(tmp_path / "cc.nii.gz"), *common], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    vol, va, inside, cc = _correlation(tmp_path / "cc.nii.gz", c, m, a)
    print("C++ command line: voxels in the mask", int(inside.sum()), "correlation with the phantom", cc)
    assert inside.sum() > 0.9 * m.sum() * a.dx * a.dy * a.dz and cc > 0.9
    assert cli.main(["-o", str(tmp_path / "py.nii.gz"), *common]) == 0
    vp, ap, ip, ccp = _correlation(tmp_path / "py.nii.gz", c, m, a)
    assert vp.shape == vol.shape and np.allclose(geo.image_to_world(ap), geo.image_to_world(va), atol=1e-6)
    assert np.array_equal(vp == -1, vol == -1) and np.abs(vp - vol).max() <= 2e-4 * np.abs(vol).max()
    # with the registrations on, the motion-free stacks stay where they are
    r = subprocess.run([build.CLI, "-o", str(tmp_path / "reg.nii.gz"), *common[:-1]], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    vr, ar, ir, ccr = _correlation(tmp_path / "reg.nii.gz", c, m, a)
    print("with registration", ccr)
    assert ccr > 0.88


@pytest.mark.gpu
def test_pvr_command_line_on_the_oblique_grid(tmp_path):
    from fetalreconstruction_amd import build
    paths, mpath, stacks, c, (m, a) = _write_case(tmp_path, 3)
    r = subprocess.run([build.PVR_CLI, "-o", str(tmp_path / "pvr.nii.gz"), "-i", *paths, "-m", mpath, "--thickness", "2.5", "2.5", "2.5",
                        "--resolution", "1.0", "--iterations", "1", "--sr_iterations", "4", "--no_registration"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    vol, va, inside, cc = _correlation(tmp_path / "pvr.nii.gz", c, m, a)
    print("PVR: voxels in the mask", int(inside.sum()), "correlation with the phantom", cc)
    assert inside.sum() > 0.8 * m.sum() * a.dx * a.dy * a.dz and cc > 0.85


def _moved_problem(prob):
    """The same problem seen from the bundled mask's frame: every matrix composed with the rigid map G that takes the
    problem's world axes to the mask's oblique axes and its origin to the mask's centre (~475 mm from the world origin)."""
    import copy
    m, a, _ = rm.load()
    G = np.eye(4)
    G[:3, 0], G[:3, 1], G[:3, 2] = a.xaxis, a.yaxis, a.zaxis
    G[:3, 3] = rm.centre(m, a)
    Gi = np.linalg.inv(G)
    q = copy.copy(prob)
    f = lambda M: np.stack([geo.to_matrix4(x) for x in M])
    M4 = lambda A: A.reshape(-1, 4, 4).astype(np.float64)
    q.slice_i2w = f(G @ M4(prob.slice_i2w))
    q.slice_w2i = f(M4(prob.slice_w2i) @ Gi)
    q.slice_t = f(G @ M4(prob.slice_t) @ Gi)
    q.slice_tinv = f(G @ M4(prob.slice_tinv) @ Gi)
    q.recon_i2w = geo.to_matrix4(G @ prob.recon_i2w.reshape(4, 4).astype(np.float64))
    q.recon_w2i = geo.to_matrix4(prob.recon_w2i.reshape(4, 4).astype(np.float64) @ Gi)
    if prob.slice_attr is not None:
        q.slice_attr = []
        for sa in prob.slice_attr:
            r = copy.copy(sa)
            r.xaxis, r.yaxis, r.zaxis = (G[:3, :3] @ np.asarray(v, np.float64) for v in (sa.xaxis, sa.yaxis, sa.zaxis))
            r.origin = (G @ np.array([*np.asarray(sa.origin, np.float64), 1.0]))[:3]
            q.slice_attr.append(r)
    q.name = prob.name + "@mask"
    return q


@pytest.mark.gpu
def test_kernel_parity_in_the_bundled_mask_frame(tiny, oracle_mod):
    """Index work stays bit-exact and the sums within tolerance when the float32 transform chain (W2I . T^-1 . reconI2W, RC.cu:223)
    carries the oblique axes and the 300-400 mm offsets of real scanner coordinates: the parity tests of test_parity_gpu.py on the
    tiny problem moved into the bundled mask's frame."""
    import tests.test_parity_gpu as TPG
    P = _moved_problem(tiny)
    assert np.abs(P.slice_i2w[:, 3]).max() > 250 and np.abs(P.slice_t.reshape(-1, 4, 4)[:, :3, :3]).max() < 1.0 + 1e-6
    TPG.test_psf_taps_are_bit_identical(P, oracle_mod, golden=False)
    # the production kernels by their mode numbers (gauss_mode 1 / fwd_mode 1 = the unit gather, back_mode 4 = the wave-owned
    # scatter, back_mode 5 = the cell-owned scatter without atomics), then the first-generation kernels
    TPG.test_gaussian_reconstruction_parity(P, oracle_mod, 1, 1)
    TPG.test_gaussian_reconstruction_parity(P, oracle_mod, 0, 0)
    TPG.test_forward_projection_parity(P, oracle_mod, 1)
    TPG.test_forward_projection_parity(P, oracle_mod, 0)
    for bm in TPG.BACK_MODES:
        TPG.test_backprojection_parity(P, oracle_mod, bm)
    TPG.test_em_steps_parity(P, oracle_mod)
    # ... and against the reference's own arithmetic (LITERAL) in this frame, where the float32 lattice is coarsest
    import tests.test_round2_gaps as TR2
    # (the literal form carries absolute slice-space positions of 300-400 mm in float32 -- an ulp of 3e-5 mm against the 1e-7
    # of the canonical residual form -- and sums them in float: its own rounding doubles the distance seen around the origin
    # (3.3e-3 on v_PSF_sums here, 1.6e-3 there); the hit sets stay identical, 0 differences in all five)
    TR2.test_hip_path_against_the_literal_oracle(P, oracle_mod, None, tol=6e-3)
    # the GPU registration of --useGPUReg (a17): sampled and blurred slices bit-exact, the same decisions at every step
    import tests.test_registration as TR
    vol = TR._analytic_volume(tiny)
    TR.test_cost_evaluation_parity(P, oracle_mod, vol=vol)
    TR.test_registration_parity(P, oracle_mod, vol=vol)
    # the patch-to-volume twins (a18): support 12, PVR constants, superpixel masks
    import tests.test_pvr as TPV
    TPV.test_pvr_taps_are_bit_identical(P, oracle_mod)
    TPV.test_pvr_psf_kernels_parity(P, oracle_mod, False, 1)
    TPV.test_pvr_psf_kernels_parity(P, oracle_mod, True, 1)
