"""Shared helpers for the parity tests."""
import numpy as np


_MEMO = {}
_SCALARS = (int, float, bool, str, type(None), np.generic)


def _snap(obj):
    out = {}
    for k, v in obj.__dict__.items():
        if isinstance(v, np.ndarray):
            out[k] = ("a", v.copy())
        elif isinstance(v, _SCALARS):
            out[k] = ("s", v)
        elif isinstance(v, list) and all(isinstance(x, _SCALARS) for x in v):
            out[k] = ("l", list(v))
    return out


def _restore(obj, snap):
    for k, (kind, v) in snap.items():
        cur = obj.__dict__.get(k)
        if kind == "a" and isinstance(cur, np.ndarray) and cur.shape == v.shape and cur.dtype == v.dtype:
            np.copyto(cur, v)                     # in place: views (recon | volw of one pair buffer) and the C side's pointers stay what they are
        else:
            setattr(obj, k, v.copy() if kind == "a" else (list(v) if kind == "l" else v))


def _memo_key(r, stage):
    """Oracle engines only (the CPU side of a parity test): everything the stages read -- the problem's arrays as the test may have edited
    them, the oracle's mode, the driver's state before the run."""
    import hashlib
    import os
    eng = getattr(r, "reconstructionGPU", None)
    if os.environ.get("SVR_TEST_NO_MEMO") or type(eng).__name__ != "OracleReconstruction" or getattr(getattr(r, "comm", None), "world", 1) != 1:
        return None
    h = hashlib.sha1()
    for a in list(getattr(eng, "_keep", [])) + [eng.slices, eng.mask] + ([eng.spx_masks] if getattr(eng, "spx_masks", None) is not None else []):
        h.update(np.ascontiguousarray(a).tobytes())
    for obj in (eng, r):
        for k, (kind, v) in sorted(_snap(obj).items()):
            h.update(k.encode())
            h.update(v.tobytes() if kind == "a" else repr(v).encode())
    return (stage, type(r).__name__, h.hexdigest())


def run_to_state(rec_driver, stage):
    """Drive irtkReconstruction up to a named stage of reconstruction.cc:930-1108.  On an oracle engine the result is remembered per session
    (same problem bytes, same oracle mode, same driver state going in => the same state coming out: the oracle is a pure function of them);
    a dozen parity tests start from the same three or four states, seconds of CPU each.  SVR_TEST_NO_MEMO=1 switches the memory off."""
    key = _memo_key(rec_driver, stage)
    if key is not None and key in _MEMO:
        eng_snap, drv_snap = _MEMO[key]
        _restore(rec_driver.reconstructionGPU, eng_snap)
        _restore(rec_driver, drv_snap)
        return
    _run_to_state(rec_driver, stage)
    if key is not None:
        _MEMO[key] = (_snap(rec_driver.reconstructionGPU), _snap(rec_driver))


def _run_to_state(rec_driver, stage):
    r = rec_driver
    r.InitializeEMValuesGPU()
    if stage == "em_init":
        return
    r.GaussianReconstructionGPU()
    if stage == "gauss":
        return
    r.SimulateSlicesGPU()
    if stage == "sim":
        return
    r.InitializeRobustStatisticsGPU()
    r.EStepGPU()
    if stage == "estep0":
        return
    r.ScaleGPU()
    if stage == "scale":
        return
    r.SuperresolutionGPU(1)
    if stage == "sr1":
        return
    r.SimulateSlicesGPU()
    r.MStepGPU(1)
    r.EStepGPU()
    if stage == "iter1":
        return
    raise ValueError(stage)


def rel_err(a, b, floor=None):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = np.max(np.abs(b)) if floor is None else floor
    return float(np.max(np.abs(a - b)) / max(scale, 1e-30))


def popcount_xor(a, b):
    return int(sum(bin(int(x) ^ int(y)).count("1") for x, y in zip(a, b)))
