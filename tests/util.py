"""Shared helpers for the parity tests."""
import numpy as np


def run_to_state(rec_driver, stage):
    """Drive irtkReconstruction up to a named stage of reconstruction.cc:930-1108."""
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
