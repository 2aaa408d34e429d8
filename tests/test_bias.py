"""Bias-correction path (unreachable from the reference CLI but part of its engine):
CorrectBias + NormaliseBias and the exp(-bias) variants of the scatter / EM kernels."""
import numpy as np
import pytest

from tests.twins.reconstruction import irtkReconstruction
from tests.util import rel_err, run_to_state


def _biased(tiny):
    """The tiny phantom with a smooth multiplicative bias field on every slice."""
    import copy
    P = copy.copy(tiny)
    ns, sy, sx = tiny.slices.shape
    yy, xx = np.meshgrid(np.linspace(-1, 1, sy), np.linspace(-1, 1, sx), indexing="ij")
    field = np.exp(0.25 * xx - 0.15 * yy)[None] * (1 + 0.05 * np.sin(np.arange(ns))[:, None, None])
    P.slices = np.where(tiny.slices > 0, tiny.slices * field, tiny.slices).astype(np.float32)
    return P


def test_oracle_bias_path_runs_and_reduces_the_field(tiny, oracle_mod):
    P = _biased(tiny)
    o = oracle_mod.OracleReconstruction(P, oracle_mod.CANON, bias_correction=True)
    d = irtkReconstruction(o, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
    d.SetSmoothingParameters(150, 0.02)
    d._disableBiasC = False
    run_to_state(d, "estep0")
    d.BiasGPU()
    act = (P.slices != -1) & (o.simweights > 0.99)
    assert np.isfinite(o.bias).all() and (np.abs(o.bias[act]) > 1e-4).any()
    for sl in range(P.ns):                                    # zero mean per slice (RC.cu:1896-1929)
        m = P.slices[sl] > -1
        if m.any() and o.bias[sl].any():
            assert abs(o.bias[sl].sum() / m.sum()) < 1e-4
    assert (o.bias[P.slices == -1] == 0).all()
    d.ScaleGPU()
    d.SuperresolutionGPU(1)
    before = o.recon.copy()
    d.NormaliseBiasGPU(0)
    assert np.isfinite(o.bias_vol).all() and np.isfinite(o.recon).all()
    assert not np.array_equal(before, o.recon)
    c = o.maskC.reshape(P.vsize[::-1])
    assert 0.1 < c[17, 17, 17] <= 1.0 and c[17, 17, 17] > c[0, 0, 0]      # blurred mask (RC.cu:1129-1157)


@pytest.mark.gpu
def test_bias_path_parity(tiny, oracle_mod):
    from fetalreconstruction_amd import engine as E
    P = _biased(tiny)
    rec = E.Reconstruction(0)
    rec.set_flags(disable_bias_correction=False)
    E.sync_gpu(rec, P)
    orc = oracle_mod.OracleReconstruction(P, oracle_mod.CANON, bias_correction=True)
    kw = dict(max_intensity=P.max_intensity, min_intensity=P.min_intensity)
    dg, do = irtkReconstruction(rec, P.ns, **kw), irtkReconstruction(orc, P.ns, **kw)
    for d in (dg, do):
        d.SetSmoothingParameters(150, 0.02)
        d._disableBiasC = False
        run_to_state(d, "estep0")
    assert rel_err(rec.debug_get(E.BUF_SMOOTH_MASK), orc.maskC) < 1e-5
    # CorrectBias on identical inputs
    for b, a in ((E.BUF_WEIGHTS, orc.weights), (E.BUF_SIMSLICES, orc.simslices), (E.BUF_SIMWEIGHTS, orc.simweights)):
        rec.debug_set(b, a)
    rec.CorrectBias(12.0, False)
    orc.CorrectBias(12.0, False)
    assert rel_err(rec.debug_get(E.BUF_BIAS), orc.bias, floor=1.0) < 2e-5
    # the exp(-bias) variants of scale / E-step / M-step / back-projection
    rec.debug_set(E.BUF_BIAS, orc.bias)
    assert rel_err(rec.CalculateScaleVector(), orc.CalculateScaleVector()) < 1e-5
    pg, po_ = rec.EStep(do._m_gpu, do._sigma_gpu, 0.9), orc.EStep(do._m_gpu, do._sigma_gpu, 0.9)
    assert rel_err(pg, po_) < 1e-5
    rec.debug_set(E.BUF_WEIGHTS, orc.weights)
    assert np.allclose(rec.MStepSums(), orc.MStepSums(), rtol=1e-5)
    rec.SuperresolutionBackproject(orc.slice_weights)
    orc.SuperresolutionBackproject(orc.slice_weights)
    assert rel_err(rec.debug_get(E.BUF_ADDON), orc.addon) < 2e-5
    assert rel_err(rec.debug_get(E.BUF_CONFIDENCE_MAP), orc.cmap) < 2e-5
    # NormaliseBias on identical inputs
    args = (do._adaptive, do._alpha, do._min_intensity, do._max_intensity, do._delta, do._lambda)
    orc.SuperresolutionUpdate(*args)
    rec.debug_set(E.BUF_RECONSTRUCTED, orc.recon)
    rec.NormaliseBias(0, 12.0)
    orc.NormaliseBias(0, 12.0)
    assert rel_err(rec.debug_get(E.BUF_BIAS_VOLUME), orc.bias_vol, floor=1.0) < 2e-5
    assert rel_err(rec.syncCPU(), orc.recon) < 2e-5


@pytest.mark.gpu
def test_full_iteration_with_bias_tracks_the_oracle(tiny, oracle_mod):
    from fetalreconstruction_amd import engine as E, host
    P = _biased(tiny)
    rec = E.Reconstruction(0)
    rec.set_flags(disable_bias_correction=False)
    E.sync_gpu(rec, P)
    orc = oracle_mod.OracleReconstruction(P, oracle_mod.CANON, bias_correction=True)
    kw = dict(max_intensity=P.max_intensity, min_intensity=P.min_intensity)
    hc = host.irtkReconstruction(rec, P.ns, **kw)                 # C++ host object
    hc.set_bias_correction(True, 12.0)
    do = irtkReconstruction(orc, P.ns, **kw)
    do._disableBiasC = False
    for d in (hc, do):
        d.SetSmoothingParameters(150, 0.02)
        d.reconstruct_iteration(2)
    st = hc.state()
    assert np.allclose(st["scale"], do._scale_gpu, rtol=2e-4)
    assert rel_err(rec.debug_get(E.BUF_BIAS), orc.bias, floor=1.0) < 2e-4
    assert rel_err(rec.syncCPU(), orc.recon) < 2e-4
