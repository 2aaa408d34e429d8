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
