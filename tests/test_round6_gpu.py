"""Round-6 regressions on the device (through the C-ABI, like every GPU test).

* the small-results arena (csrc/svr_hip.hip down_queue / down_reserve): a batch of queued device -> host copies whose LATER items are
  larger than the arena's first size.  Round 5 refused to grow an arena with copies in flight, so the slice-level EM's fetch -- four
  vectors of the GLOBAL slice count -- failed with SVR_E_STATE on every rank that held a fraction of the slices, and on one GPU above
  ~990 slices.  The multi-process tests used 48 slices and never reached it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _many_slices():
    # 1120 slices of 12 x 12 pixels: 16 ns + 128 > 12 ns + 4096 bytes (what the arena's first size was)
    from tests import test_two_ranks_one_gpu as T
    return T._problem("many")


def test_the_slice_level_em_comes_down_on_a_run_of_a_thousand_slices():
    from fetalreconstruction_amd import engine as E, host
    P = _many_slices()
    assert P.ns > 1000
    rec = E.Reconstruction(0)
    E.sync_gpu(rec, P)
    d = host.irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
    d.SetSmoothingParameters(150, 0.02)
    d.reconstruct_iteration(2)                  # the second outer iteration's InitializeEMValues pulls the device's slice-level state
    st = d.state()                              # ... and so does svrh_get_state
    assert st["scale"].shape == (P.ns,) and np.isfinite(st["scale"]).all() and np.isfinite(st["slice_weight"]).all()
    assert np.isfinite(rec.syncCPU()[rec.syncCPU() != -1]).all()
    rec.close()


@pytest.mark.timeout(900)
def test_a_rank_with_an_eighth_of_a_thousand_slices_fetches_the_global_vectors():
    """world 4 on the one GPU (processes over gloo, tests/test_two_ranks_one_gpu.py's launcher): every rank holds 280 of 1120 slices; the
    vectors the slice-level EM hands down have 1120 entries"""
    import os
    import tempfile
    from tests import test_two_ranks_one_gpu as T
    from fetalreconstruction_amd import engine as E, host
    P = _many_slices()
    rec = E.Reconstruction(0)
    E.sync_gpu(rec, P)
    ref = host.irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
    ref.SetSmoothingParameters(150, 0.02)
    ref.reconstruct_iteration(3)
    v_ref, s_ref = rec.syncCPU().copy(), ref.state()
    rec.close()
    with tempfile.TemporaryDirectory() as d:
        T._spawn(4, True, True, d, seed="many", kind="svr")
        rr = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(4)]
    for r1 in rr[1:]:
        for k in ("recon", "scale", "sw", "pot", "em"):
            assert np.array_equal(rr[0][k], r1[k], equal_nan=True), k
    order = rr[0]["order"]
    assert np.abs(rr[0]["recon"] - v_ref).max() <= 2e-5 * np.abs(v_ref).max()
    assert np.allclose(rr[0]["scale"], s_ref["scale"][order], rtol=1e-5) and np.allclose(rr[0]["sw"], s_ref["slice_weight"][order], atol=1e-4)


def test_whichever_psf_pass_comes_first_writes_the_table(tiny):
    """coeff_lazy: after a new slice geometry (or svr_set_option coeff_invalidate) the first evaluating PSF pass on the cell path writes the coefficient
    table -- pass 2 of the Gaussian reconstruction in the reconstruction loop, the scatter in bench.py's steps, the gather if it comes first -- and
    every order gives the bits of evaluating every tap in every pass.  A pixel whose factors are both zero in the writing scatter (weight 0) is
    evaluated and stored all the same: a later pass with other weights reads its units."""
    from fetalreconstruction_amd import engine as E
    rng = np.random.default_rng(4)
    ones = np.ones(tiny.ns, np.float32)
    w1 = np.where(tiny.slices != -1, rng.uniform(0.2, 1.0, tiny.slices.shape) * (rng.uniform(0, 1, tiny.slices.shape) > 0.5), 0).astype(np.float32)   # half the weights 0
    w2 = np.where(tiny.slices != -1, rng.uniform(0.2, 1.0, tiny.slices.shape), 0).astype(np.float32)
    sim = np.where(tiny.slices > 0, tiny.slices * rng.uniform(0.8, 1.2, tiny.slices.shape), 0).astype(np.float32)
    V = rng.uniform(0.5, 1.5, tiny.nvox).astype(np.float32)

    def run(order):
        rec = E.Reconstruction(0)
        E.sync_gpu(rec, tiny)
        rec.timer_enable(True)
        if order == "evaluate":
            rec.set_option("coeff_table", 0)
        rec.UpdateScaleVector(ones, ones)
        rec.InitializeEMValues()
        rec.GaussianReconstruction()
        out = {"volw": rec.getVolWeights().copy(), "recon": rec.syncCPU().copy()}
        if order != "evaluate":
            assert rec.get_option("coeff_valid") == 1                      # pass 2 wrote it
            if order in ("scatter", "gather"):
                rec.set_option("coeff_invalidate", 1)
        rec.debug_set(E.BUF_RECONSTRUCTED, V)
        rec.debug_set(E.BUF_SIMSLICES, sim)

        def scatter(w):
            rec.debug_set(E.BUF_WEIGHTS, w)
            rec.SuperresolutionBackproject(ones)
            return rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy()

        def gather():
            rec.SimulateSlices()
            g = tuple(rec.debug_get(b).copy() for b in (E.BUF_SIMSLICES, E.BUF_SIMWEIGHTS, E.BUF_SIMINSIDE))
            rec.debug_set(E.BUF_SIMSLICES, sim)
            return g

        if order == "gather":
            out["g1"] = gather()
            out["s1"] = scatter(w1)
        else:
            out["s1"] = scatter(w1)                                         # order "scatter": this one writes the table, half its pixels at weight 0
            out["g1"] = gather()
        out["s2"] = scatter(w2)                                             # ... and this one reads the units of the pixels that had weight 0
        out["g2"] = gather()
        tm = rec.timers()
        out["stores"] = (tm["backproject_store"][1], tm["forward_store"][1])
        out["valid"] = rec.get_option("coeff_valid")
        rec.close()
        return out

    ref = run("evaluate")
    for order, stores in (("gauss", (0, 0)), ("scatter", (1, 0)), ("gather", (0, 1))):
        o = run(order)
        assert o["valid"] == 1 and o["stores"] == stores, (order, o["stores"])
        for k in ("volw", "recon"):
            assert np.array_equal(o[k], ref[k], equal_nan=True), (order, k)
        for k in ("s1", "g1", "s2", "g2"):
            for a, b in zip(o[k], ref[k]):
                assert np.array_equal(a, b, equal_nan=True), (order, k)
