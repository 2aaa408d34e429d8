"""The WHOLE of P4 (BASELINE.json configs[1]: 280 slices, 1.15 M active pixels, 117 x 109 x 90 voxels on the reference's bundled
mask frame) through the oracle, dealt over the host threads, against the HIP path -- every pixel and every voxel, not a sample:

  * CANON (the sequence the device implements): the `sume` gate, voxcount, siminside and the scatter's hit sets exactly;
    v_PSF_sums to 1e-6; volw, the reconstructed volume, simulated slices / weights, addon and cmap to the float-sum tolerance;
  * LITERAL (the reference's own operation sequence with libm): hit-set symmetric differences bounded as on the small phantom,
    values within TOL_LITERAL -- asserted at BASELINE size, not only on `tiny`.

The slices are dealt to one oracle instance per host thread (its own partial volume, like slice-sharded ranks; the C calls
release the GIL); partial volumes are added in double.  About a minute per mode on the gpurun box's 16 threads.

The same for PVR4 (BASELINE configs[2]: the patch-to-volume kernels, support 12, 5.27 M patch pixels) by default, and for the 8-stack
workloads on request (SVR_FULL_WORKLOADS).  Recorded one-off runs on the MI355X box (round 4, final kernels): S8 CANON -- all 10 607 524
PSF pixels, every hit set identical (0 differences of 10.6 M pixels / 9.93 M voxels), v_PSF_sums identical, sums within 6.3e-7 of the
buffer's maximum; S8 LITERAL -- every hit set identical, share beyond 3e-3 max <= 4.0e-4, relative L2 <= 2.6e-3; PVR8spx CANON -- 12 237 835 PSF
pixels, every hit set identical (31.6 M voxels), v_PSF_sums identical, sums within 3.3e-6."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from fetalreconstruction_amd import phantom, workloads
from tests.util import rel_err

pytestmark = pytest.mark.gpu

TOL_SUM = 2e-5          # tests/test_parity_gpu.py
TOL_LITERAL = 3e-3      # tests/test_round2_gaps.py
# observed on the MI355X box (psf_sums / volw / recon / sim / simw / addon / cmap): share beyond TOL_LITERAL 1.0e-4 / 6.4e-5 / 0 / 1.7e-5 /
# 4.3e-6 / 1.7e-4 / 6.4e-5; relative L2 1.9e-4 / 1.3e-4 / 1.4e-5 / 5.7e-5 / 2.5e-5 / 1.7e-3 (addon: a residual, sums that cancel) /
# 1.4e-4; worst element 7.4e-2 / 1.9e-2 / 2.0e-3 / 2.1e-2 / 2.2e-2 / 1.5e-1 / 2.9e-2 of the buffer's maximum
MAX_SHARE_BEYOND_TOL = 1e-3
MAX_REL_L2 = 5e-3
# (round 6: no cap of "half the buffer's range" on the worst element any more -- every element beyond TOL_LITERAL must lie on a pixel, or in the
# footprint of a pixel, where the canonical and the literal walk of the epsilon-skip part, and v_PSF_sums there is bounded by the flipped taps' mass)


def _deal(prob, parts):
    act = (prob.slices != -1).reshape(prob.ns, -1).sum(1)
    bins, load = [[] for _ in range(parts)], np.zeros(parts)
    for i in np.argsort(-act, kind="stable"):
        t = int(np.argmin(load))
        bins[t].append(int(i))
        load[t] += act[i]
    return [np.array(sorted(b)) for b in bins if b]


def _oracle_pass(prob, oracle_mod, mode, V, weights, simslices, threads, pvr=False):
    """Gaussian reconstruction, forward projection of V, scatter with the given per-pixel state: every slice, one oracle
    instance per thread -> full-size slice-grid arrays and summed volumes"""
    parts = _deal(prob, threads)
    ns, sy, sx = prob.slices.shape
    nv = prob.nvox
    out = dict(psf_sums=np.zeros((ns, sy, sx), np.float32), voxcount=np.zeros((ns, sy, sx), np.int32), sim=np.zeros((ns, sy, sx), np.float32),
               simw=np.zeros((ns, sy, sx), np.float32), inside=np.zeros((ns, sy, sx), np.uint8),
               recon=np.zeros(nv, np.float64), volw=np.zeros(nv, np.float64), addon=np.zeros(nv, np.float64), cmap=np.zeros(nv, np.float64))

    def run(idx):
        sub = phantom.sub_problem(prob, 0, 0, select=idx)
        spx = getattr(prob, "spx_masks", None)
        o = oracle_mod.OracleReconstruction(sub, mode, pvr=pvr, spx_masks=None if spx is None else np.ascontiguousarray(spx[idx]))
        ones = np.ones(sub.ns, np.float32)
        o.UpdateScaleVector(ones, ones)
        o.InitializeEMValues()
        o.GaussianReconstructionLocal()                        # recon | volw: this instance's partial sums (no equalize)
        part = dict(recon=o.recon.astype(np.float64), volw=o.volw.astype(np.float64))
        o.recon[...] = V
        o.SimulateSlices()
        o.weights[...] = weights[idx]
        sim_fwd = o.simslices.copy()
        o.simslices[...] = simslices[idx]
        o.SuperresolutionBackproject(ones)
        part.update(addon=o.addon.astype(np.float64), cmap=o.cmap.astype(np.float64))
        return idx, o.psf_sums.copy(), o.voxcount.copy(), sim_fwd, o.simweights.copy(), o.siminside.copy(), part

    with ThreadPoolExecutor(len(parts)) as pool:
        for idx, ps, vc, sim, sw, si, part in pool.map(run, parts):
            out["psf_sums"][idx], out["voxcount"][idx], out["sim"][idx], out["simw"][idx], out["inside"][idx] = ps, vc, sim, sw, si
            for k in ("recon", "volw", "addon", "cmap"):
                out[k] += part[k]
    return out


# By default P4 (BASELINE configs[1]) and PVR4 (configs[2]: 5 149 patches of 32 x 32, 5.27 M pixels, support 12: a minute per mode).
# SVR_FULL_WORKLOADS=S8 or PVR8spx: the same comparison on the 8-stack workloads (S8: 10.6 M active pixels, 20 M voxels -- 4.4 minutes per
# mode on 16 host threads: one-off runs, recorded in DESIGN 6, not part of the default suite)
@pytest.mark.parametrize("workload", __import__("os").environ.get("SVR_FULL_WORKLOADS", "P4,PVR4").split(","))
@pytest.mark.parametrize("mode_name", ["CANON", "LITERAL"])
def test_every_pixel_and_voxel_of_p4_against_the_oracle(mode_name, workload, oracle_mod, capsys):
    from fetalreconstruction_amd import engine as E
    if workload == "PVR4" and mode_name == "LITERAL" and "SVR_FULL_WORKLOADS" not in __import__("os").environ:
        pytest.skip("a minute of the default suite: runs with SVR_FULL_WORKLOADS=PVR4 (recorded: every hit set identical, share beyond 3e-3 max <= 5.6e-5)")
    if workload == "P4" and mode_name == "LITERAL" and "SVR_FULL_WORKLOADS" not in __import__("os").environ:
        # round 5: the suite is held under 12 minutes on the MI355X box (the driver's step has 20).  The LITERAL comparison stays in the suite on the
        # tiny problem and in the bundled mask's frame (tests/test_round2_gaps.py, tests/test_real_geometry.py); at full size it runs with
        # SVR_FULL_WORKLOADS=P4 (recorded in round 4: share of elements beyond 3e-3 max <= 1.7e-4, relative L2 <= 1.7e-3) and S8 / PVR8spx are
        # committed logs (profiles/r05_full_workload_*.txt)
        pytest.skip("half a minute of the default suite: runs with SVR_FULL_WORKLOADS=P4")
    P = workloads.get(workload)
    ns, sy, sx = P.slices.shape
    rng = np.random.default_rng(17)
    V = rng.uniform(0.5, 1.5, P.nvox).astype(np.float32)
    weights = np.where(P.slices != -1, rng.uniform(0.2, 1.0, P.slices.shape), 0).astype(np.float32)
    simslices = np.where(P.slices > 0, P.slices * rng.uniform(0.8, 1.2, P.slices.shape), 0).astype(np.float32)

    pvr = workload.startswith("PVR")                       # (patches as the units: support 12, the patch-to-volume constants)
    rec = E.Reconstruction(0)
    if pvr:
        rec.set_option("pvr", 1)
        E.sync_gpu(rec, P, quality_factor=1.0)
        if getattr(P, "spx_masks", None) is not None:
            rec.set_spx_masks(P.spx_masks)
    else:
        E.sync_gpu(rec, P)
    ones = np.ones(P.ns, np.float32)
    rec.UpdateScaleVector(ones, ones)
    rec.InitializeEMValues()
    g = {}
    rec.GaussianReconstruction()
    g["volw"] = rec.getVolWeights().copy()
    g["recon"] = rec.syncCPU().copy()                      # equalized: recon / volw
    g["psf_sums"] = rec.debug_get(E.BUF_PSF_SUMS).copy()
    g["voxcount"] = rec.debug_get(E.BUF_VOXEL_COUNT).copy()
    rec.debug_set(E.BUF_RECONSTRUCTED, V)
    rec.SimulateSlices()
    g["sim"], g["simw"], g["inside"] = (rec.debug_get(b).copy() for b in (E.BUF_SIMSLICES, E.BUF_SIMWEIGHTS, E.BUF_SIMINSIDE))
    rec.debug_set(E.BUF_WEIGHTS, weights)
    rec.debug_set(E.BUF_SIMSLICES, simslices)
    rec.SuperresolutionBackproject(ones)
    g["addon"], g["cmap"] = rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy()
    threads = max(1, min(int(E.load_library().svr_host_threads()), 32))
    rec.close()

    mode = getattr(oracle_mod, mode_name)
    o = _oracle_pass(P, oracle_mod, mode, V, weights, simslices, threads, pvr)
    o["recon"] = np.where(o["volw"] != 0, o["recon"] / np.where(o["volw"] != 0, o["volw"], 1), o["recon"])      # equalizeVol RC.cu:2312-2327
    va = int(((P.slices != -1) & (o["psf_sums"] != 0)).sum())
    sym = {k: (int(((g[k] != 0) != (o[k] != 0)).sum()), int((o[k] != 0).sum())) for k in ("psf_sums", "voxcount", "inside", "volw", "cmap")}
    errs = {k: rel_err(g[k], o[k]) for k in ("psf_sums", "volw", "recon", "sim", "simw", "addon", "cmap")}
    with capsys.disabled():
        print(f"\n[{workload} whole, HIP vs {mode_name} oracle on {threads} threads] {va} PSF pixels of {int((P.slices != -1).sum())}; hit-set differences "
              + ", ".join(f"{k}: {a}/{b}" for k, (a, b) in sym.items()) + "; max |diff| / max |ref| "
              + ", ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    assert va > 1_000_000
    if mode_name == "CANON":
        for k, (a, b) in sym.items():
            assert a == 0, (k, a, b)                                         # index work: bit-exact
        assert np.array_equal(g["voxcount"], o["voxcount"])
        assert np.allclose(g["psf_sums"], o["psf_sums"], rtol=1e-6, atol=0, equal_nan=True)
        for k in ("volw", "recon", "sim", "simw", "addon", "cmap"):
            assert errs[k] < TOL_SUM, (k, errs[k])
    else:
        # A flipped skip decision moves a whole tap in or out of ONE pixel's sums (and can re-phase the rest of its row): over a
        # million pixels the maximum norm is set by a handful of such pixels (the census bounds one pixel's change, tests/census.py),
        # so at this size the statement is about the distribution: the share of elements beyond TOL_LITERAL, the relative L2 error,
        # and a cap on the worst element
        stats = {}
        for k in errs:
            d = np.abs(g[k].astype(np.float64).reshape(-1) - o[k].astype(np.float64).reshape(-1))
            ref = np.abs(o[k].astype(np.float64).reshape(-1))
            nz = ref > 0
            stats[k] = (float((d[nz] > TOL_LITERAL * ref.max()).mean()), float(np.sqrt((d ** 2).sum() / max((ref ** 2).sum(), 1e-300))), errs[k])
        with capsys.disabled():
            print(f"[{workload} whole, LITERAL] share of elements beyond %.0e of the maximum / relative L2 / worst element: " % TOL_LITERAL
                  + ", ".join(f"{k} {a:.1e} / {b:.1e} / {c:.1e}" for k, (a, b, c) in stats.items()))
        for k, (a, b) in sym.items():
            assert a <= max(2, b // 2000), (k, a, b)                         # a flipped tap adds or drops a voxel at the rim of a footprint
        for k, (share, l2, worst) in stats.items():
            assert share < MAX_SHARE_BEYOND_TOL and l2 < MAX_REL_L2, (k, share, l2, worst)
        # attribution: the census of EVERY pixel (literal against canonical walk), then every outlier on / under a flipped pixel
        from tests.test_bench_size_oracle import _flips
        sub_all = oracle_mod.OracleReconstruction(P, mode, pvr=pvr, spx_masks=getattr(P, "spx_masks", None))
        pixels = np.argwhere(P.slices != -1)
        flips, open_, mass = _flips(sub_all, pixels, threads)
        flipped = np.zeros(P.slices.shape, bool)
        flipped[tuple(pixels[flips > 0].T)] = True
        fmass = np.zeros(P.slices.shape, np.float32)
        fmass[tuple(pixels.T)] = mass
        vz, vy, vx = P.mask.shape
        near = np.zeros((vz, vy, vx), bool)
        S = 12 if pvr else 16
        lo_, hi_ = (S - 1) // 2, S - 1 - (S - 1) // 2
        for sl, py, px in pixels[flips > 0]:
            c = sub_all.tap_census(int(sl), int(px), int(py))[3]
            cx, cy, cz = int(c[0]), int(c[1]), int(c[2])
            near[max(cz - lo_, 0):max(cz + hi_ + 1, 0), max(cy - lo_, 0):max(cy + hi_ + 1, 0), max(cx - lo_, 0):max(cx + hi_ + 1, 0)] = True
        n_out = {}
        for k in ("psf_sums", "sim", "simw"):
            d = np.abs(np.nan_to_num(g[k].astype(np.float64)) - np.nan_to_num(o[k].astype(np.float64)))
            out_ = d > TOL_LITERAL * np.abs(np.nan_to_num(o[k])).max()
            n_out[k] = int(out_.sum())
            assert not np.any(out_ & ~flipped), (k, "an element beyond the tolerance on a pixel whose two walks agree")
            if k == "psf_sums":
                assert np.all(d[out_] <= 1.05 * fmass[out_] + TOL_LITERAL * np.abs(np.nan_to_num(o[k])).max()), "v_PSF_sums moved by more than its flipped taps carry"
        for k in ("volw", "recon", "addon", "cmap"):
            d = np.abs(g[k].astype(np.float64).reshape(-1) - o[k].astype(np.float64).reshape(-1))
            out_ = d > TOL_LITERAL * np.abs(o[k]).max()
            n_out[k] = int(out_.sum())
            assert not np.any(out_ & ~near.reshape(-1)), (k, "a voxel beyond the tolerance outside every flipped pixel's footprint")
        with capsys.disabled():
            print(f"[{workload} whole, LITERAL] {int((flips > 0).sum())} of {len(pixels)} pixels with a flipped decision ({int(((flips == 0) & (open_ > 0)).sum())} more with one the "
                  f"reference's own build leaves open); elements beyond the tolerance, every one on / under a flipped pixel: " + ", ".join(f"{k} {v}" for k, v in n_out.items()))


# ---- a whole outer iteration at BASELINE size -----------------------------------------------------------------------------------
class _ThreadGroup:
    """ranks = threads of this process (the oracle engines release the GIL in their C calls): what the launcher's communicator is to
    the slice-sharded host loop (tests/twins/reconstruction.py), with sums taken in rank order"""

    def __init__(self, world):
        import threading
        self.world = world
        self.slots = [None] * world
        self.barrier = threading.Barrier(world)

    def gather(self, rank, a):
        self.slots[rank] = a
        self.barrier.wait(timeout=900)                         # (a rank that failed breaks the barrier for the others instead of hanging them)
        out = list(self.slots)
        self.barrier.wait(timeout=900)
        return out


class _ThreadComm:
    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, rank, group.world

    def _all(self, a):
        return self.g.gather(self.rank, np.array(a, np.float64))

    def allreduce_sum(self, a):
        parts = self._all(a)
        out = parts[0].copy()
        for p in parts[1:]:
            out = out + p
        return out

    def allreduce_min(self, a):
        return np.min(np.stack(self._all(a)), axis=0)

    def allreduce_max(self, a):
        return np.max(np.stack(self._all(a)), axis=0)

    def allgather_slices(self, local, counts):
        parts = self.g.gather(self.rank, np.asarray(local, np.float32).copy())
        return np.concatenate([p[:c] for p, c in zip(parts, counts)])

    def allreduce_volume_pair(self, engine, which):
        buf = engine.recon_volw if which == 0 else engine.addon_cmap
        parts = self.g.gather(self.rank, buf.astype(np.float64))
        tot = parts[0].copy()
        for p in parts[1:]:
            tot += p
        buf[...] = tot.astype(buf.dtype)


@pytest.mark.timeout(1800)
def test_a_whole_outer_iteration_of_p4_tracks_the_oracle(oracle_mod, capsys):
    """BASELINE configs[1] end to end: InitializeEMValues, Gaussian reconstruction, forward projection, robust-statistics
    initialisation, E-step and TWO super-resolution iterations (Scale, back-projection, the volume update, forward projection,
    M-step, E-step) -- the C++ host on the HIP engine against the Python mirror of the host loop on the oracle (CANON), the oracle's
    slices sharded over the host threads as ranks of an in-process group.  Each side evolves its own state; the tolerances are those of
    the tiny problem's `test_full_iteration_tracks_the_oracle`."""
    from concurrent.futures import ThreadPoolExecutor
    from fetalreconstruction_amd import engine as E, host
    from fetalreconstruction_amd.sharding import shard_slices
    from tests.twins.reconstruction import irtkReconstruction
    wl = __import__("os").environ.get("SVR_OUTER_WORKLOAD", "P4")      # (S8: a one-off run of ten minutes, recorded in DESIGN 6)
    P = workloads.get(wl)
    rec = E.Reconstruction(0)
    E.sync_gpu(rec, P)
    dg = host.irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
    dg.SetSmoothingParameters(150, 0.02)
    dg.reconstruct_iteration(2)
    sg = dg.state()
    vol_g = rec.syncCPU().copy()
    threads = max(2, min(int(E.load_library().svr_host_threads()), 16))
    rec.close()

    act = (P.slices != -1).reshape(P.ns, -1).sum(1)
    ranges = shard_slices(act, threads)
    group = _ThreadGroup(threads)

    def rank_main(r):
        lo, hi = ranges[r]
        eng = oracle_mod.OracleReconstruction(phantom.sub_problem(P, lo, hi), oracle_mod.CANON)
        drv = irtkReconstruction(eng, P.ns, (lo, hi), _ThreadComm(group, r), P.max_intensity, P.min_intensity)
        drv.SetSmoothingParameters(150, 0.02)
        drv.reconstruct_iteration(2)
        return eng.recon.copy(), np.asarray(drv._scale_gpu).copy(), np.asarray(drv._slice_weight_gpu).copy(), (drv._sigma_gpu, drv._mix_gpu, drv._m_gpu)

    def guarded(r):
        try:
            return rank_main(r)
        except BaseException:
            group.barrier.abort()
            raise

    with ThreadPoolExecutor(threads) as pool:
        outs = list(pool.map(guarded, range(threads)))
    vol_o, scale_o, sw_o, em_o = outs[0]
    for o in outs[1:]:
        assert np.array_equal(o[0], vol_o)                                 # every rank ends with the same volume
    err = rel_err(vol_g, vol_o)
    with capsys.disabled():
        print(f"\n[{wl}, one outer iteration with 2 SR iterations, HIP (C++ host) vs oracle on {threads} thread-ranks] volume max |diff| / max |ref| {err:.1e}; "
              f"sigma {sg['sigma']:.6g} / {em_o[0]:.6g}, mix {sg['mix']:.6g} / {em_o[1]:.6g}, m {sg['m']:.6g} / {em_o[2]:.6g}; "
              f"slices at weight < 0.5: {int((np.asarray(sg['slice_weight']) < 0.5).sum())} / {int((sw_o < 0.5).sum())}")
    assert np.allclose(sg["scale"], scale_o, rtol=1e-4)
    assert np.allclose(sg["slice_weight"], sw_o, atol=1e-3)
    assert np.allclose([sg["sigma"], sg["mix"], sg["m"]], em_o, rtol=1e-4)
    assert np.array_equal(vol_g == -1, vol_o == -1)
    assert err < 1e-4


@pytest.mark.timeout(1800)
def test_a_whole_outer_iteration_of_pvr4_tracks_the_oracle(oracle_mod, capsys):
    """BASELINE configs[2] end to end, as above with patches as the units: the C++ patch-based host (csrc/pvr_host.cpp) on the HIP engine
    against the Python mirror (tests/twins/pvr.py) on the oracle, 5 149 patches sharded over the host threads; Gaussian reconstruction,
    robust statistics and two super-resolution iterations.  Tolerances of `test_pvr_loop_parity` (the small phantom), except for the scale
    factor of a patch ONE OF WHOSE PIXELS CROSSES A GATE between the two sides.  A patch's factor is sum w s sim / sum w s^2 over its pixels with
    simulated weight > 0.99 (patchBasedRobustStatistics_gpu.cu:672-745); the simulated weight of a pixel is a float sum of up to 1728 taps and
    agrees between the device and the oracle to 3e-7, so a pixel whose weight is 0.99 on one side and 0.99000007 on the other is in one sum and
    not in the other -- one of ~590 pixels: the factor moves by a few 1e-4 (profiles/r05_pvr4_scale_diag.txt, tools/pvr4_scale_diag.py: patches
    838 and 839, one such pixel each; over the COMMON pixels the two sides agree to 1e-4 and 5e-5).  The reference's own float atomics would put
    such a pixel on either side from run to run.  So: every patch beyond 1e-4 must have a gate-crossing pixel in the final state, and stay
    within 3 / (pixels in its sum); everything else holds the small phantom's tolerances (volume: observed 7.4e-6)."""
    from concurrent.futures import ThreadPoolExecutor
    from fetalreconstruction_amd import engine as E, host
    from fetalreconstruction_amd.sharding import shard_slices
    from tests.twins import pvr
    P = workloads.get("PVR4")
    rec = E.Reconstruction(0)
    rec.set_option("pvr", 1)
    E.sync_gpu(rec, P, quality_factor=1.0)
    dg = host.irtkPatchBasedReconstruction(rec, P.patches_per_stack, P.min_intensity, P.max_intensity)
    dg.reconstruct_iteration(2)
    sg = dg.state()
    vol_g = rec.syncCPU().copy()
    simw_g = rec.debug_get(E.BUF_SIMWEIGHTS).reshape(P.slices.shape).copy()
    threads = max(2, min(int(E.load_library().svr_host_threads()), 16))
    rec.close()

    ranges = shard_slices((P.slices > 0).reshape(P.ns, -1).sum(1), threads)
    group = _ThreadGroup(threads)

    def rank_main(r):
        lo, hi = ranges[r]
        eng = oracle_mod.OracleReconstruction(phantom.sub_problem(P, lo, hi), oracle_mod.CANON, pvr=True)
        drv = pvr.irtkPatchBasedReconstruction(eng, P.patches_per_stack, P.min_intensity, P.max_intensity, patch_range=(lo, hi), comm=_ThreadComm(group, r))
        drv.reconstruct_iteration(2)
        return eng.recon.copy(), np.asarray(drv.scale).copy(), np.asarray(drv.patch_weight).copy(), (drv.m_sigma_gpu, drv.m_mix_gpu, drv.m_m_gpu), eng.simweights.copy()

    def guarded(r):
        try:
            return rank_main(r)
        except BaseException:
            group.barrier.abort()
            raise

    with ThreadPoolExecutor(threads) as pool:
        outs = list(pool.map(guarded, range(threads)))
    vol_o, scale_o, pw_o, em_o = outs[0][:4]
    for o in outs[1:]:
        assert np.array_equal(o[0], vol_o)
    simw_o = np.concatenate([o[4] for o in outs])
    data = P.slices > 0
    gate_g, gate_o = (simw_g > 0.99) & data, (simw_o > 0.99) & data
    crossing = (gate_g != gate_o).reshape(P.ns, -1).sum(1)                  # pixels on different sides of the gate, per patch (final state)
    in_sum = np.maximum(gate_o.reshape(P.ns, -1).sum(1), 1)
    assert np.abs(simw_g - simw_o)[data].max() < 2e-6
    err = rel_err(vol_g, vol_o)
    em_g = [sg["m_sigma_gpu"], sg["m_mix_gpu"], sg["m_m_gpu"]]
    with capsys.disabled():
        print(f"\n[PVR4, one outer iteration with 2 SR iterations, HIP (C++ host) vs oracle on {threads} thread-ranks] volume max |diff| / max |ref| {err:.1e}; "
              f"sigma {em_g[0]:.6g} / {em_o[0]:.6g}, mix {em_g[1]:.6g} / {em_o[1]:.6g}, m {em_g[2]:.6g} / {em_o[2]:.6g}")
    bad = ~np.isclose(sg["scale"], scale_o, rtol=1e-4)
    with capsys.disabled():
        if bad.any():
            i = np.flatnonzero(bad)
            print(f"  scale differs at {len(i)} of {len(bad)} patches, e.g. {i[:6]}: HIP {np.asarray(sg['scale'])[i[:6]]} oracle {scale_o[i[:6]]}; "
                  f"weights there HIP {np.asarray(sg['patch_weight'])[i[:6]]} oracle {pw_o[i[:6]]}; pixels in the sum {in_sum[i[:6]]}, gate-crossing pixels {crossing[i[:6]]} "
                  f"(of {int(crossing.sum())} in {int((crossing > 0).sum())} patches overall)")
    assert np.allclose(em_g, em_o, rtol=1e-4)
    # every patch beyond 1e-4 has a pixel on different sides of the `simulated weight > 0.99` gate, and stays within 3 / n of its sum
    assert bad.mean() < 2e-3 and (crossing[bad] > 0).all(), np.flatnonzero(bad & (crossing == 0))
    assert (np.abs(np.asarray(sg["scale"]) / scale_o - 1.0)[bad] <= 3.0 / in_sum[bad]).all()
    assert np.allclose(sg["patch_weight"], pw_o, atol=1e-3)
    assert err < 1e-4
