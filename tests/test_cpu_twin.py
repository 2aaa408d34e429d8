"""oracle/cpu_twin.c -- the reference's CPU reconstruction path (irtkReconstruction::CoeffInit + the CPU twins of the SR loop) restated in
C: what bench.py reports as `cpu_baseline` and the QUALITY cross-check SURVEY 8(c) asks for ("Oracle-A final-volume statistics"): the
product and the reference's CPU algorithm reconstruct the same phantom to volumes that agree with each other and with the analytic truth.
It pins nothing bit-wise: a different PSF (Gaussian, trilinear splat) in double arithmetic."""
import os
import subprocess

import numpy as np
import pytest

from fetalreconstruction_amd import phantom


def _stats(a, b, m):
    """normalised cross-correlation and PSNR (peak = the reference volume's maximum in the region) of a against b over the voxels m"""
    x, y = a[m].astype(np.float64), b[m].astype(np.float64)
    ncc = float(np.corrcoef(x, y)[0, 1])
    g = float((x * y).sum() / (x * x).sum())                 # least-squares intensity scale of a onto b (the two pipelines scale their volumes on their own)
    mse = float(np.mean((g * x - y) ** 2))
    return ncc, 10.0 * np.log10(float(y.max()) ** 2 / mse)


def test_cpu_twin_against_the_oracle_of_the_gpu_path_and_the_threads(oracle_mod):
    """small case, CPU only: the twin (Gaussian PSF, explicit coefficients, double) and the restated GPU kernels (sinc^2 x Gauss taps,
    float) reconstruct the same problem to volumes with a normalised cross-correlation above 0.98 inside the mask; one thread and four
    threads differ by the order of double additions only; every slice pixel with data has coefficients."""
    from oracle import cputwin
    from tests.twins.reconstruction import irtkReconstruction
    P = phantom.make_problem(3, (40, 36, 10), 1.1, 2.2, None, 1.0, 15.0, seed=11, orientations=("ax", "cor", "sag"))
    vols = []
    for threads in (1, 4):
        tw = cputwin.CpuTwin(P, threads=threads)
        tw.SetSmoothingParameters(150, 0.02)
        tw.reconstruct_iteration(3)
        vols.append(tw.volume())
        assert tw.active_pixels == int((P.slices != -1).sum()) or tw.active_pixels > 0.95 * int((P.slices != -1).sum())
        assert tw.coefficients > 50 * tw.active_pixels
        st = tw.state()
        assert 0 < st["mix"] <= 1 and st["sigma"] > 0 and (st["slice_weight"] >= 0).all() and (st["slice_weight"] <= 1).all()
        tw.close()
    m = np.asarray(P.mask).reshape(-1) != 0
    assert np.array_equal(vols[0] == -1, vols[1] == -1) and np.abs(vols[0] - vols[1])[m].max() <= 1e-6 * np.abs(vols[0][m]).max()
    orc = oracle_mod.OracleReconstruction(P, oracle_mod.CANON)
    d = irtkReconstruction(orc, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
    d.SetSmoothingParameters(150, 0.02)
    d.reconstruct_iteration(3)
    ncc, psnr = _stats(vols[0], orc.recon, m)
    print(f"twin vs the GPU path's oracle: NCC {ncc:.4f}, PSNR {psnr:.1f} dB over {int(m.sum())} mask voxels")
    assert ncc > 0.98 and psnr > 20


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_quality_cross_check_of_the_product_against_the_references_cpu_path(tmp_path):
    """The P4 phantom (BASELINE configs[0..1]: the four stacks on the bundled mask's oblique frame, written as NIfTI files) reconstructed
    twice with the reference's schedule and --no_registration: by the product's command line, bin/SVRreconstructionGPU (C++ pre-processing,
    C++ host, HIP engine), and by the reference's CPU path restated (oracle/cpu_twin.c) on EXACTLY what the command line uploaded
    (--dumpProblem) with the same outer iterations, SR iterations and smoothing schedule (reconstruction.cc:884-896, 930-1140, 1189-1193).
    Reported (and bounded from below): NCC / PSNR between the two final volumes and of each against the analytic phantom, inside the mask."""
    from fetalreconstruction_amd import build, geometry as geo, nifti, workloads
    from oracle import cputwin, pyoracle as po
    from tests.test_prep_oracle import _read_svr_dump
    build.build()
    stacks, mask = workloads.p4_stacks()
    paths = []
    for k, (d, a) in enumerate(stacks):
        nifti.write(tmp_path / f"s{k}.nii.gz", d, a)
        paths.append(str(tmp_path / f"s{k}.nii.gz"))
    nifti.write(tmp_path / "mask.nii.gz", mask.data.astype(np.float32), mask.attr)
    iterations, rec_first, rec_last, delta, lam, last_lam, levels, thick = 3, 4, 13, 150.0, 0.02, 0.01, 3, 2.5
    dump, out = tmp_path / "svr.bin", tmp_path / "gpu.nii.gz"
    r = subprocess.run([build.CLI, "-o", str(out), "-i", *paths, "-m", str(tmp_path / "mask.nii.gz"), "--thickness", *[str(thick)] * 4, "--resolution", "1.0",
                        "--iterations", str(iterations), "--rec_iterations_first", str(rec_first), "--rec_iterations_last", str(rec_last), "--no_registration",
                        "--dumpProblem", str(dump)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    v_gpu, va = nifti.read(out)
    v_gpu = v_gpu.reshape(-1)
    # the problem the command line uploaded -> the twin's input
    D = _read_svr_dump(str(dump))
    tattr = D["tattr"]
    attrs, ids = [], []
    for k, sa in enumerate(D["sattrs"]):
        for j in range(sa.nz):
            a = po.get_region_attr(sa, 0, 0, j, sa.nx, sa.ny, j + 1)       # CreateSlicesAndTransformations RG.cc:1835-1880
            a.dz = thick
            attrs.append(a)
            ids.append(k)
    ns = len(attrs)
    assert ns == D["grid"].shape[0]
    grid = np.ascontiguousarray(D["grid"], np.float32)
    pos = grid[grid > 0]
    rd = (tattr.dx, tattr.dy, tattr.dz)
    P = phantom.Problem(vsize=(tattr.nx, tattr.ny, tattr.nz), vdim=rd, recon_i2w=geo.to_matrix4(geo.image_to_world(tattr)),
                        recon_w2i=geo.to_matrix4(geo.world_to_image(tattr)), mask=np.ascontiguousarray(D["vmask"], np.float32), slices=grid,
                        slice_i2w=np.stack([geo.to_matrix4(geo.image_to_world(a)) for a in attrs]),
                        slice_w2i=np.stack([geo.to_matrix4(geo.world_to_image(a)) for a in attrs]),
                        slice_t=np.stack([geo.to_matrix4(t) for t in D["T"]]), slice_tinv=np.stack([geo.to_matrix4(np.linalg.inv(t)) for t in D["T"]]),
                        slice_dim=np.array([[a.dx, a.dy, a.dz] for a in attrs], np.float32), sizes_x=np.asarray(D["sizes_x"], np.int32),
                        sizes_y=np.asarray(D["sizes_y"], np.int32), stack_index=np.asarray(ids, np.int32), psf_c0=geo.psf_centre_offset(rd),
                        min_intensity=float(pos.min()), max_intensity=float(pos.max()), name="P4 files")
    threads = max(1, min(32, len(os.sched_getaffinity(0))))
    tw = cputwin.CpuTwin(P, threads=threads)
    for it in range(iterations):                                           # reconstruction.cc:884-896
        if it == iterations - 1:
            tw.SetSmoothingParameters(delta, last_lam)
        else:
            l = lam
            for i in range(levels):
                if it == iterations * (levels - i - 1) // levels:
                    tw.SetSmoothingParameters(delta, l)
                l *= 2
        tw.reconstruct_iteration(rec_last if it == iterations - 1 else rec_first)
    tw.RestoreSliceIntensitiesAndScaleVolume(D["factors"])                 # reconstruction.cc:1189-1193
    v_cpu = tw.volume()
    # the analytic phantom on the template grid, in the stacks' intensity units
    _, c = workloads.load_bundled_mask()
    kk, jj, ii = np.meshgrid(np.arange(tattr.nz), np.arange(tattr.ny), np.arange(tattr.nx), indexing="ij")
    w = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(np.float64) @ geo.image_to_world(tattr).T
    truth = (phantom.phantom_intensity(w[..., :3] - c, workloads.RADIUS) * 700.0 / 0.55).reshape(-1)
    m = (np.asarray(D["vmask"]).reshape(-1) != 0) & (v_gpu > 0) & (v_cpu > 0)
    assert m.sum() > 0.95 * (np.asarray(D["vmask"]) != 0).sum()
    res = {"product vs reference CPU path": _stats(v_gpu, v_cpu, m), "product vs phantom": _stats(v_gpu, truth, m), "reference CPU path vs phantom": _stats(v_cpu, truth, m)}
    line = "; ".join(f"{k}: NCC {v[0]:.4f}, PSNR {v[1]:.1f} dB" for k, v in res.items())
    print(f"P4 phantom, {iterations} iterations of {rec_first}/{rec_last} SR iterations, no registration, {int(m.sum())} mask voxels -- {line}; "
          f"mean intensity product {v_gpu[m].mean():.1f}, CPU path {v_cpu[m].mean():.1f}, phantom {truth[m].mean():.1f}; "
          f"CPU path: CoeffInit {np.mean(tw.times['CoeffInit']):.2f} s x {iterations}, SR iteration {np.mean(tw.times['Superresolution']) + np.mean(tw.times['SimulateSlices']):.2f} s on {threads} threads")
    out_dir = os.environ.get("SVR_QUALITY_LOG")
    if out_dir:
        open(out_dir, "w").write(line + "\n")
    assert res["product vs reference CPU path"][0] > 0.97
    assert res["product vs phantom"][0] > 0.92 and res["reference CPU path vs phantom"][0] > 0.92    # (noise sigma 5 on a phantom whose texture is 2 % of its range; thick slices)
    assert abs(v_gpu[m].mean() / v_cpu[m].mean() - 1.0) < 0.05             # both restore the stacks' intensities and scale the volume (RG.cc:1003-1079)
