"""ctypes binding of oracle/cpu_twin.c: the REFERENCE'S CPU RECONSTRUCTION PATH restated in plain C (irtkReconstruction::CoeffInit,
GaussianReconstruction, SimulateSlices, EStep, Scale, Superresolution, MStep ... of irtkReconstructionGPU.cc; Gaussian PSF, trilinear
splat, explicit coefficient lists, double arithmetic).  TEST INFRASTRUCTURE: bench.py's cpu_baseline leg times it, tests/ cross-check
final-volume quality against it.  It pins nothing (see the C file's header) and is not the bit-parity oracle of the GPU path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libcpu_twin.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "cpu_twin.c")):
            subprocess.check_call(["make", "-C", HERE, "libcpu_twin.so"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(LIB)
        _lib.twin_create.restype = C.c_void_p
        _lib.twin_coefficients.restype = C.c_uint64
        _lib.twin_active_pixels.restype = C.c_uint64
        for name in ("twin_destroy", "twin_set_smoothing_parameters", "twin_set_force_excluded", "twin_coeff_init", "twin_initialize_em_values",
                     "twin_gaussian_reconstruction", "twin_simulate_slices", "twin_initialize_robust_statistics", "twin_estep", "twin_scale",
                     "twin_superresolution", "twin_mask_volume", "twin_get_volume", "twin_get_state", "twin_restore_and_scale_volume"):
            getattr(_lib, name).restype = None
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class CpuTwin:
    """the `--useCPU` branch of reconstruction.cc:930-1140 on one phantom.Problem"""

    def __init__(self, prob, threads=1):
        L = lib()
        vx, vy, vz = (int(v) for v in prob.vsize)
        ns, sy, sx = prob.slices.shape
        self.prob, self.threads = prob, int(threads)
        keep = [np.ascontiguousarray(prob.mask, np.float32), np.ascontiguousarray(prob.slices, np.float32),
                np.ascontiguousarray(prob.sizes_x, np.int32), np.ascontiguousarray(prob.sizes_y, np.int32),
                np.ascontiguousarray(prob.slice_i2w, np.float32), np.ascontiguousarray(prob.slice_t, np.float32),
                np.ascontiguousarray(prob.recon_w2i, np.float32), np.ascontiguousarray(prob.slice_dim, np.float32)]
        self._h = C.c_void_p(L.twin_create(vx, vy, vz, C.c_double(float(prob.vdim[0])), _p(keep[0]), ns, sx, sy, _p(keep[1]), _p(keep[2]), _p(keep[3]),
                                           _p(keep[4]), _p(keep[5]), _p(keep[6]), _p(keep[7]), C.c_double(float(prob.min_intensity)),
                                           C.c_double(float(prob.max_intensity)), int(threads)))
        self.times = {}

    def close(self):
        if getattr(self, "_h", None):
            lib().twin_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _timed(self, key, fn, *a):
        t0 = time.perf_counter()
        r = fn(self._h, *a)
        self.times.setdefault(key, []).append(time.perf_counter() - t0)
        return r

    def SetSmoothingParameters(self, delta, lam):
        lib().twin_set_smoothing_parameters(self._h, C.c_double(delta), C.c_double(lam))

    def CoeffInit(self):
        self._timed("CoeffInit", lib().twin_coeff_init)

    def InitializeEMValues(self):
        self._timed("InitializeEMValues", lib().twin_initialize_em_values)

    def GaussianReconstruction(self):
        self._timed("GaussianReconstruction", lib().twin_gaussian_reconstruction)

    def SimulateSlices(self):
        self._timed("SimulateSlices", lib().twin_simulate_slices)

    def InitializeRobustStatistics(self):
        self._timed("InitializeRobustStatistics", lib().twin_initialize_robust_statistics)

    def EStep(self):
        self._timed("EStep", lib().twin_estep)

    def Scale(self):
        self._timed("Scale", lib().twin_scale)

    def Superresolution(self, it):
        self._timed("Superresolution", lib().twin_superresolution, int(it))

    def MStep(self, it):
        if self._timed("MStep", lib().twin_mstep, int(it)):
            raise RuntimeError("MStep: mix = 0 (the reference exits here, RG.cc:4246-4249)")

    def MaskVolume(self):
        lib().twin_mask_volume(self._h)

    def RestoreSliceIntensitiesAndScaleVolume(self, stack_factor):
        """RG.cc:1003-1024 + 1034-1079: what reconstruction.cc:1189-1193 does after the last iteration"""
        f = np.ascontiguousarray(stack_factor, np.float32)
        si = np.ascontiguousarray(self.prob.stack_index, np.int32)
        lib().twin_restore_and_scale_volume(self._h, _p(f), _p(si))

    def sr_iteration(self, i):
        """reconstruction.cc:1013-1108 with useCPU, bias correction off"""
        self.Scale()
        self.Superresolution(i + 1)
        self.SimulateSlices()
        self.MStep(i + 1)
        self.EStep()

    def preamble(self):
        """reconstruction.cc:930-1001 with useCPU"""
        self.InitializeEMValues()
        self.CoeffInit()
        self.GaussianReconstruction()
        self.SimulateSlices()
        self.InitializeRobustStatistics()
        self.EStep()

    def reconstruct_iteration(self, rec_iterations):
        self.preamble()
        for i in range(rec_iterations):
            self.sr_iteration(i)
        self.MaskVolume()

    @property
    def coefficients(self):
        return int(lib().twin_coefficients(self._h))

    @property
    def active_pixels(self):
        return int(lib().twin_active_pixels(self._h))

    def volume(self):
        out = np.empty(int(np.prod(self.prob.vsize)), np.float32)
        lib().twin_get_volume(self._h, _p(out))
        return out

    def state(self):
        ns = self.prob.slices.shape[0]
        sc, sw, s8 = np.zeros(ns), np.zeros(ns), np.zeros(8)
        lib().twin_get_state(self._h, _p(sc), _p(sw), _p(s8))
        names = ("sigma", "mix", "m", "mean_s", "mean_s2", "sigma_s", "sigma_s2", "mix_s")
        return dict(scale=sc, slice_weight=sw, **{k: float(v) for k, v in zip(names, s8)})
