"""ctypes loader for the CPU oracle (oracle/svr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by fetalreconstruction_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

LITERAL, CANON = 0, 1


class Geom(C.Structure):
    _fields_ = [
        ("vx", C.c_int), ("vy", C.c_int), ("vz", C.c_int),
        ("vdim", C.c_float * 3),
        ("reconI2W", C.c_float * 16), ("reconW2I", C.c_float * 16),
        ("sx", C.c_int), ("sy", C.c_int), ("ns", C.c_int),
        ("sliceI2W", C.c_void_p), ("sliceW2I", C.c_void_p),
        ("T", C.c_void_p), ("Tinv", C.c_void_p), ("sliceDim", C.c_void_p),
        ("psf_c0", C.c_float * 3),
        ("psf_mode", C.c_int),
        ("bias2D", C.c_void_p),
        ("pvr", C.c_int),
        ("spx_mask", C.c_void_p),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libsvr_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("svr_oracle.c", "reg_oracle.c", "prep_oracle.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsvr_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_gaussian_reconstruction.restype = C.c_int
        _LIB.orc_tap_census.restype = C.c_int
        _LIB.orc_initialize_robust_statistics.restype = C.c_float
        _LIB.orc_scale_volume.restype = C.c_float
        _LIB.orc_ncc_evaluate.restype = C.c_double
        _LIB.orc_reg_gauss_kernel.restype = C.c_int
        _LIB.orc_reg_tex3d.restype = C.c_float
        _LIB.orc_cc_patch.restype = C.c_float
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleReconstruction:
    """CPU stand-in for `class Reconstruction` with the same method names as
    fetalreconstruction_amd.engine.Reconstruction, so the host driver
    (tests.twins.reconstruction.irtkReconstruction) can run on either and the parity
    tests compare buffer by buffer.  Includes the reference's host-glue quirks that live below the
    boundary (e.g. the one-call lag of the device scale vector, RC.cu:3195,3238)."""

    def __init__(self, prob, mode=CANON, bias_correction=False, pvr=False, spx_masks=None):
        self.prob = prob
        self.pvr = bool(pvr)
        self.spx_masks = None if spx_masks is None else np.ascontiguousarray(spx_masks, np.uint8)
        self.mode = mode
        self.bias_correction = bool(bias_correction)
        # private copies: tests edit these through `_keep` and must not touch the (session-wide) problem
        self._keep = [_f32(prob.slice_i2w).copy(), _f32(prob.slice_w2i).copy(), _f32(prob.slice_t).copy(),
                      _f32(prob.slice_tinv).copy(), _f32(prob.slice_dim).copy()]
        g = Geom()
        g.vx, g.vy, g.vz = prob.vsize
        g.vdim[:] = prob.vdim
        g.reconI2W[:] = _f32(prob.recon_i2w).tolist()
        g.reconW2I[:] = _f32(prob.recon_w2i).tolist()
        ns, sy, sx = prob.slices.shape
        g.sx, g.sy, g.ns = sx, sy, ns
        g.sliceI2W, g.sliceW2I, g.T, g.Tinv, g.sliceDim = [a.ctypes.data for a in self._keep]
        g.psf_c0[:] = _f32(prob.psf_c0).tolist()
        g.psf_mode = mode
        g.pvr = int(self.pvr)
        g.spx_mask = self.spx_masks.ctypes.data if self.spx_masks is not None else None
        self.g = g
        self.vsize = tuple(prob.vsize)
        self.sgrid = (ns, sy, sx)
        self.slices = _f32(prob.slices).copy()
        self.mask = _f32(prob.mask).reshape(-1)
        nv = prob.nvox
        shp = self.slices.shape
        self.recon_volw = np.zeros(2 * nv, np.float32)
        self.recon, self.volw = self.recon_volw[:nv], self.recon_volw[nv:]
        self.addon_cmap = np.zeros(2 * nv, np.float32)
        self.addon, self.cmap = self.addon_cmap[:nv], self.addon_cmap[nv:]
        self.psf_sums = np.zeros(shp, np.float32)
        self.voxcount = np.zeros(shp, np.int32)
        self.weights = np.zeros(shp, np.float32)
        self.simslices = np.zeros(shp, np.float32)
        self.simweights = np.zeros(shp, np.float32)
        self.siminside = np.zeros(shp, np.uint8)
        self.d_scales = np.ones(ns, np.float32)
        self.h_scales = np.ones(ns, np.float32)
        self.slice_weights = np.ones(ns, np.float32)
        self.bias = None
        if self.bias_correction:                       # _disableBiasC == false
            self.bias = np.zeros(shp, np.float32)
            g.bias2D = self.bias.ctypes.data
            self.wb = np.zeros(shp, np.float32)
            self.wr = np.zeros(shp, np.float32)
            self.buffer = np.zeros(shp, np.float32)
            self.bias_vol = np.zeros(nv, np.float32)
            self.volume_weights = np.zeros(nv, np.float32)
            self.maskC = np.zeros(nv, np.float32)
            vx, vy, vz = prob.vsize
            lib().orc_smooth_mask(vx, vy, vz, (C.c_float * 3)(*[float(d) for d in prob.vdim]), _p(self.mask),
                                  C.c_float(12.0), _p(self.maskC))

    def _b(self):
        return _p(self.bias) if self.bias is not None else None

    # ---- state -------------------------------------------------------------------------
    def UpdateScaleVector(self, scales, slice_weights):
        self.h_scales = _f32(scales).copy()
        self.d_scales = _f32(scales).copy()
        self.slice_weights = _f32(slice_weights).copy()

    def UpdateSliceWeights(self, slice_weights):
        self.slice_weights = _f32(slice_weights).copy()

    def SetSliceMatrices(self, slice_transforms, inv_slice_transforms, i2w_init, w2i_init, i2w, w2i, recon_i2w, recon_w2i):
        """UpdateGPUTranformationMatrices: new slice transformations (the geometry struct points at the kept arrays)"""
        self._keep[2][:] = _f32(slice_transforms).reshape(self._keep[2].shape)
        self._keep[3][:] = _f32(inv_slice_transforms).reshape(self._keep[3].shape)

    def register_patches(self, ri2w, mo, invmo, transformations, levels=3, steps=4, iterations=20):
        """the engine's svr_pvr_register_patches on the oracle's patches and current reconstruction"""
        vx, vy, vz = self.vsize
        return pvr_register_patches(self.slices, ri2w, mo, invmo, transformations, self.prob.recon_w2i, self.recon.reshape(vz, vy, vx),
                                    float(self.prob.vdim[0]), levels, steps, iterations)

    def syncCPU(self):
        return self.recon.copy()

    # ---- Reconstruction::GaussianReconstruction (reconstruction_cuda2.cu:2329-2493) --------
    def GaussianReconstructionLocal(self):
        if not self.pvr:     # PVR: ReconVolume::reset() leaves the patch buffers alone (reconVolume.cuh:93-98)
            for a in (self.weights, self.simslices, self.simweights):   # RC.cu:2402-2409
                a[...] = 0
            self.siminside[...] = 0
        self._gauss_n = lib().orc_gaussian_reconstruction(
            C.byref(self.g), _p(self.slices), _p(self.d_scales), _p(self.mask), _p(self.recon), _p(self.volw),
            _p(self.psf_sums), _p(self.voxcount))

    def GaussianReconstructionFinish(self):
        lib().orc_equalize(C.c_size_t(self.recon.size), _p(self.recon), _p(self.volw))
        return self._gauss_n

    def GaussianReconstruction(self):
        self.GaussianReconstructionLocal()
        return [self.GaussianReconstructionFinish()]

    def SimulateSlices(self):
        inside = np.zeros(self.g.ns, np.uint8)
        lib().orc_simulate_slices(C.byref(self.g), _p(self.slices), _p(self.psf_sums), _p(self.recon), _p(self.mask),
                                  _p(self.simslices), _p(self.simweights), _p(self.siminside), _p(inside))
        return inside.astype(bool)

    def sample_pixels(self, flat_indices, recon):
        """Pass 1 of the Gaussian reconstruction and the forward projection of `recon` for the listed slice-grid pixels only
        (orc_sample_pixels) -> sume, keep, sim, weight, inside."""
        idx = np.ascontiguousarray(flat_indices, np.uint32)
        n = len(idx)
        out = (np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint8))
        lib().orc_sample_pixels(C.byref(self.g), _p(self.slices), _p(_f32(recon).reshape(-1)), _p(self.mask), _p(idx), n,
                                *[_p(o) for o in out])
        return out

    def SuperresolutionBackproject(self, slice_weight=None):
        if slice_weight is not None:
            self.UpdateSliceWeights(slice_weight)
        lib().orc_superresolution_backproject(C.byref(self.g), _p(self.slices), _p(self.weights), _p(self.simslices),
                                              _p(self.slice_weights), _p(self.d_scales), _p(self.mask),
                                              _p(self.psf_sums), _p(self.addon), _p(self.cmap))

    def SuperresolutionUpdate(self, adaptive, alpha, min_i, max_i, delta, lam):
        vx, vy, vz = self.prob.vsize
        original = self.recon.copy()
        lib().orc_regularization_prep(vx, vy, vz, int(bool(adaptive)), C.c_float(alpha), C.c_float(min_i),
                                      C.c_float(max_i), _p(self.recon), _p(self.addon), _p(self.cmap))
        lib().orc_regularization(vx, vy, vz, C.c_float(delta), C.c_float(alpha), C.c_float(lam),
                                 _p(self.recon), _p(original), _p(self.cmap))

    def Superresolution(self, it, slice_weight, adaptive, alpha, min_i, max_i, delta, lam,
                        global_bias_correction=False, sigma_bias=12.0, low_intensity_cutoff=0.01):
        self.SuperresolutionBackproject(slice_weight)
        self.SuperresolutionUpdate(adaptive, alpha, min_i, max_i, delta, lam)

    def InitializeEMValues(self):
        f = lib().orc_initialize_em_values_pvr if self.pvr else lib().orc_initialize_em_values
        f(C.c_size_t(self.slices.size), _p(self.slices), _p(self.weights))
        if self.bias is not None:
            self.bias[...] = 0                          # RC.cu:3305-3309

    # ---- bias path (RC.cu:1837-1942, 2519-2652) ---------------------------------------------
    def CorrectBias(self, sigma_bias, global_bias_correction=False):
        lib().orc_correct_bias(self.g.sx, self.g.sy, self.g.ns, _p(self.slices), _p(self.bias), _p(self.weights),
                               _p(self.simweights), _p(self.simslices), _p(self.d_scales), _p(self._keep[4]),
                               C.c_float(sigma_bias), int(bool(global_bias_correction)), _p(self.wb), _p(self.wr),
                               _p(self.buffer))

    def NormaliseBias(self, it, sigma_bias):
        lib().orc_normalise_bias(C.byref(self.g), _p(self.slices), _p(self.d_scales), _p(self.mask),
                                 _p(self.psf_sums), _p(self.volw), _p(self.maskC), C.c_float(sigma_bias),
                                 _p(self.bias_vol), _p(self.volume_weights), _p(self.recon))

    def RobustStatisticsSums(self):
        s, n = C.c_double(0), C.c_double(0)
        lib().orc_initialize_robust_statistics(C.c_size_t(self.slices.size), _p(self.slices), _p(self.siminside),
                                               _p(self.simslices), _p(self.simweights), C.byref(s), C.byref(n))
        return np.array([s.value, n.value])

    def InitializeRobustStatistics(self):
        s = self.RobustStatisticsSums()
        return float(np.float32(s[0]) / np.float32(s[1]))

    def EStep(self, m, sigma, mix):
        pot = np.zeros(self.g.ns, np.float32)
        if self.pvr:
            lib().orc_estep_pvr(self.g.sx, self.g.sy, self.g.ns, _p(self.slices), _p(self.simslices),
                                _p(self.simweights), _p(self.d_scales), C.c_float(m), C.c_float(sigma),
                                C.c_float(mix), _p(self.weights), _p(pot))
            return pot
        lib().orc_estep(self.g.sx, self.g.sy, self.g.ns, _p(self.slices), _p(self.simslices), _p(self.simweights),
                        _p(self.d_scales), C.c_float(m), C.c_float(sigma), C.c_float(mix), _p(self.weights), _p(pot),
                        self._b())
        return pot

    def MStepSums(self):
        out5 = np.zeros(5, np.float64)
        lib().orc_mstep_sums(self.g.sx, self.g.sy, self.g.ns, _p(self.slices), _p(self.weights), _p(self.simslices),
                             _p(self.simweights), _p(self.h_scales), _p(out5), self._b())   # h_scales: RC.cu:3093
        return out5

    def MStep(self, it, step, sigma, mix):
        out5 = self.MStepSums()
        s, mx, m = C.c_float(sigma), C.c_float(mix), C.c_float(0)
        lib().orc_mstep_finish(_p(out5), int(it), C.c_float(step), C.byref(s), C.byref(mx), C.byref(m))
        return s.value, mx.value, m.value

    def CalculateScaleVector(self):
        sc = np.zeros(self.g.ns, np.float32)
        lib().orc_calculate_scale_vector(self.g.sx, self.g.sy, self.g.ns, _p(self.slices), _p(self.weights),
                                         _p(self.simslices), _p(self.simweights), _p(sc), self._b())
        self.d_scales = self.h_scales.copy()   # RC.cu:3238 uploads the PREVIOUS h_scales ...
        self.h_scales = sc.copy()              # ... and RC.cu:3195 then replaces it
        return sc

    def maskVolume(self):
        lib().orc_mask_volume(C.c_size_t(self.recon.size), _p(self.recon), _p(self.mask))

    def ScaleVolume(self):
        return float(lib().orc_scale_volume(self.g.sx, self.g.sy, self.g.ns, _p(self.slices), _p(self.weights),
                                            _p(self.simslices), _p(self.simweights), _p(self.slice_weights),
                                            C.c_size_t(self.recon.size), _p(self.recon)))

    def RestoreSliceIntensities(self, stack_factors, stack_index):
        f = _f32(stack_factors)
        i = np.ascontiguousarray(stack_index, np.int32)
        lib().orc_restore_slice_intensities(self.g.sx, self.g.sy, self.g.ns, _p(self.slices), _p(f), _p(i))

    # ---- gloo test hooks (see reconstruction.TorchComm.allreduce_volume_pair) ----------------
    def volume_pair_tensor(self, which):
        import torch
        return torch.from_numpy(self.recon_volw if which == 0 else self.addon_cmap)

    def volume_pair_commit(self, which, t):
        pass   # the tensor aliases the numpy buffer

    def debug_get(self, which):
        return {0: self.recon, 1: self.volw, 2: self.addon, 3: self.cmap, 4: self.mask, 10: self.slices,
                11: self.weights, 12: self.simslices, 13: self.simweights, 14: self.psf_sums,
                15: self.bias, 20: self.siminside, 21: self.voxcount, 5: getattr(self, "bias_vol", None),
                6: getattr(self, "maskC", None)}[which]

    # ---- per-pixel probes ------------------------------------------------------------------
    def tap_census(self, sl, px, py, with_vals=False):
        bits = np.zeros(64, np.uint64)
        vals = np.zeros(4096, np.float32) if with_vals else None
        c = np.zeros(3, np.float32)
        n = lib().orc_tap_census(C.byref(self.g), int(sl), int(px), int(py), _p(bits),
                                 _p(vals) if with_vals else None, _p(c))
        return n, bits, vals, c

    def fastmath_census(self, pixels):
        """orc_fastmath_census over pixels [(slice, py, px), ...] (np.argwhere order) -> dict; see svr_oracle.c"""
        pix = np.ascontiguousarray(np.asarray(pixels, np.int32)[:, [0, 2, 1]])
        out = np.zeros(16, np.float64)
        lib().orc_fastmath_census(C.byref(self.g), int(len(pix)), _p(pix), _p(out))
        names = ("taps", "kept", "uncertain", "pixels_uncertain", "mass_kept", "mass_uncertain", "sume_rel_max", "env_max", "env_mean",
                 "canon_outside", "canon_dmax", "canon_over_env_max", "canon_flips_explained", "canon_flips")
        return {k: float(v) for k, v in zip(names, out)}

    def flip_pixels(self, pixels):
        """orc_flip_pixels over pixels [(slice, py, px), ...] (np.argwhere order) -> flips, open decisions, flipped PSF mass per pixel (literal
        against canonical walk; whatever this instance's own mode is)"""
        pix = np.ascontiguousarray(np.asarray(pixels, np.int32).reshape(-1, 3)[:, [0, 2, 1]])
        n = len(pix)
        flips, open_, mass = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float32)
        if n:
            lib().orc_flip_pixels(C.byref(self.g), n, _p(pix), _p(flips), _p(open_), _p(mass))
        return flips, open_, mass

    def psf_values(self, sl, px, py):
        v = np.zeros(4096, np.float32)
        lib().orc_psf_values(C.byref(self.g), int(sl), int(px), int(py), _p(v))
        return v


def sub_problem(prob, lo, hi):
    from fetalreconstruction_amd.phantom import sub_problem as _sp
    return _sp(prob, lo, hi)


def host_estep(slice_potential, slice_weight, scale, force_excluded, small_slices, step, state5):
    """irtkReconstruction::EStepGPU host part (irtkReconstructionGPU.cc:3203-3438)."""
    pot = _f32(slice_potential).copy()
    w = _f32(slice_weight).copy()
    sc = _f32(scale)
    fe = np.ascontiguousarray(force_excluded, np.int32)
    ss = np.ascontiguousarray(small_slices, np.int32)
    st = _f32(state5).copy()
    lib().orc_host_estep(len(pot), _p(pot), _p(w), _p(sc), _p(fe), len(fe), _p(ss), len(ss), C.c_double(step), _p(st))
    return pot, w, st


def ncc_evaluate(target, M, source):
    """irtkImageRigidRegistrationWithPadding::Evaluate on one (target slice, source volume, matrix)."""
    t = np.ascontiguousarray(target, np.int16)
    if t.ndim == 2:
        t = t[None]
    src = np.ascontiguousarray(source, np.int16)
    m = np.ascontiguousarray(M, np.float64).reshape(16)
    sums = np.zeros(6, np.float64)
    tz, ty, tx = t.shape
    vz, vy, vx = src.shape
    v = lib().orc_ncc_evaluate(_p(t), tx, ty, tz, _p(m), _p(src), vx, vy, vz, _p(sums))
    return float(v), sums


class RegState(C.Structure):
    """struct orc_reg (oracle/reg_oracle.c)"""
    _fields_ = [
        ("W", C.c_int), ("H", C.c_int), ("slices", C.c_int),
        ("vx", C.c_int), ("vy", C.c_int), ("vz", C.c_int),
        ("volume", C.c_void_p), ("reconW2I", C.c_void_p), ("ofs", C.c_void_p),
        ("resampled", C.c_void_p), ("resampled_float", C.c_void_p), ("reg", C.c_void_p), ("tmp", C.c_void_p),
        ("matrices", C.c_void_p), ("matrices_orig", C.c_void_p), ("similarities", C.c_void_p),
        ("gradient", C.c_void_p), ("active", C.c_void_p), ("active2", C.c_void_p), ("active_prev", C.c_void_p),
        ("temp_float", C.c_void_p), ("temp_int", C.c_void_p),
        ("levels", C.c_int), ("steps", C.c_int), ("iterations", C.c_int), ("epsilon", C.c_float),
        ("blurring", C.c_float * 8), ("length_of_steps", C.c_float * 8),
        ("reg_dbg", C.c_void_p),
    ]


class OracleRegistration:
    """CPU stand-in for the registration entry points of `class Reconstruction`
    (initRegStorageVolumes / FillRegSlices / updateResampledSlicesI2W / prepareSliceToVolumeReg /
    registerSlicesToVolume, RC.cuh:326-338) with the method names of
    fetalreconstruction_amd.engine.Reconstruction."""

    def __init__(self, vsize, vdim, recon_w2i):
        self.vsize = tuple(int(v) for v in vsize)
        self.vdim = float(vdim)
        self.w2i = _f32(np.asarray(recon_w2i).reshape(16))
        self.st = RegState()
        self.st.vx, self.st.vy, self.st.vz = self.vsize
        self.st.reconW2I = self.w2i.ctypes.data
        self.counters = np.zeros(4, np.int64)

    def _bind(self, name, arr):
        setattr(self, "_" + name, arr)
        setattr(self.st, name, arr.ctypes.data)

    def initRegStorageVolumes(self, W, H, ns):
        st = self.st
        st.W, st.H, st.slices = int(W), int(H), int(ns)
        n = int(ns) * int(H) * int(W)
        for nm in ("resampled", "resampled_float", "reg", "tmp"):
            self._bind(nm, np.zeros(n, np.float32))
        self._bind("reg_dbg", np.full(3 * n, -1, np.float32))
        self._bind("matrices", np.zeros(16 * ns, np.float32))
        self._bind("matrices_orig", np.zeros(16 * ns, np.float32))
        self._bind("similarities", np.zeros(5 * ns, np.float32))
        self._bind("gradient", np.zeros(7 * ns, np.float32))
        for nm in ("active", "active2", "active_prev"):
            self._bind(nm, np.arange(ns, dtype=np.int32))
        self._bind("temp_float", np.zeros(6 * ns, np.float32))
        self._bind("temp_int", np.zeros(2 * ns, np.int32))
        self._bind("ofs", np.zeros(16 * ns, np.float32))

    def FillRegSlices(self, sdata, slices_resampled_i2w=None):
        self._resampled[:] = _f32(sdata).reshape(-1)
        self._resampled_float[:] = self._resampled

    def updateResampledSlicesI2W(self, ofs):
        self._ofs[:] = _f32(ofs).reshape(-1)

    def prepareSliceToVolumeReg(self, volume):
        self._bind("volume", _f32(volume).reshape(-1).copy())
        lib().orc_reg_prepare(C.byref(self.st), C.c_float(self.vdim))

    def set_schedule(self, levels=None, steps=None, iterations=None):
        if levels is not None:
            self.st.levels = int(levels)
        if steps is not None:
            self.st.steps = int(steps)
        if iterations is not None:
            self.st.iterations = int(iterations)

    def registerSlicesToVolume(self, transf):
        t = _f32(transf).reshape(-1).copy()
        lib().orc_reg_register(C.byref(self.st), _p(t), _p(self.counters))
        return t.reshape(-1, 4, 4)

    def evaluate_costs(self, transf, level, active=None):
        """One evaluateCostsMultipleSlices(…, 0, 1, 1) on the blurred targets of `level` for the given
        matrices and active list; returns (similarity[slices], blurred sampled slices [3][a][H][W])."""
        st = self.st
        ns = st.slices
        self._matrices[:] = _f32(transf).reshape(-1)
        act = np.arange(ns, dtype=np.int32) if active is None else np.asarray(active, np.int32)
        self._active[:len(act)] = act
        st.active = self._active.ctypes.data
        lib().orc_reg_begin_level(C.byref(st), C.c_int(level))
        self._similarities[:] = 0
        lib().orc_reg_evaluate_costs(C.byref(st), C.c_int(len(act)), C.c_int(level), C.c_float(st.blurring[level]),
                                     C.c_int(0), C.c_int(1), C.c_int(1))
        dbg = self._reg_dbg.reshape(3, ns, st.H, st.W)[:, :len(act)].copy()
        return self._similarities[:ns].copy(), dbg


def reg_gauss_kernel(sigma):
    half = np.zeros(32, np.float32)
    k = lib().orc_reg_gauss_kernel(C.c_float(sigma), _p(half))
    return k, half[:(k + 1) // 2].copy()


def reg_adjust(m, part, step):
    out = np.zeros(16, np.float32)
    lib().orc_reg_adjust(_p(_f32(m).reshape(16)), _p(out), C.c_int(part), C.c_float(step))
    return out.reshape(4, 4)


def reg_gradient_step(m, g, step):
    out = _f32(m).reshape(16).copy()
    lib().orc_reg_gradient_step(_p(out), _p(_f32(g)), C.c_float(step))
    return out.reshape(4, 4)


def pvr_register_patches(patches, ri2w, mo, invmo, transformations, recon_w2i, volume, recon_dim_x, levels=3, steps=4, iterations=20):
    """PatchBased2D3DRegistration_gpu2<T>::run restated (reg_oracle.c: orc_pvr_register_patches) -> (T, Tinv, counters3)."""
    b = _f32(patches)
    n, py, px = b.shape
    r, m, mi = (_f32(a).reshape(n, 16) for a in (ri2w, mo, invmo))
    t = _f32(transformations).reshape(n, 16).copy()
    ti = np.zeros((n, 16), np.float32)
    w, v = _f32(recon_w2i).reshape(16), _f32(volume)
    vz, vy, vx = v.shape
    c = np.zeros(3, np.int64)
    lib().orc_pvr_register_patches(_p(b), px, py, n, _p(r), _p(m), _p(mi), _p(t), _p(ti), _p(w), _p(v), vx, vy, vz,
                                   C.c_float(recon_dim_x), int(levels), int(steps), int(iterations), _p(c))
    return t, ti, c


def pvr_blur_patches(patches, sigma):
    b = _f32(patches)
    n, py, px = b.shape
    out = np.zeros_like(b)
    lib().orc_pvr_blur_patches(_p(b), _p(out), px, py, n, C.c_float(sigma))
    return out


def pvr_params(matrix):
    m = _f32(matrix).reshape(16)
    p, r = np.zeros(6, np.float32), np.zeros(16, np.float32)
    lib().orc_pvr_params(_p(m), _p(p), _p(r))
    return p, r.reshape(4, 4)


def cc_patches(buffers, ri2w, tmats, recon_w2i, volume, level):
    """computeCCpatch for every patch i with its own matrix: buffers [n][pY][pX], ri2w / tmats [n][16],
    volume [vz][vy][vx].  Returns (ncc[n] float32, sums [n][6] = n, sum a, sum b, sum a^2, sum b^2, sum ab)."""
    b = _f32(buffers)
    n, py, px = b.shape
    r, t, w, v = _f32(ri2w).reshape(n, 16), _f32(tmats).reshape(n, 16), _f32(recon_w2i).reshape(16), _f32(volume)
    vz, vy, vx = v.shape
    out = np.zeros(n, np.float32)
    sums = np.zeros((n, 6), np.float32)
    for i in range(n):
        out[i] = lib().orc_cc_patch(_p(b[i]), px, py, _p(r[i]), _p(t[i]), _p(w), _p(v), vx, vy, vz, int(level),
                                    _p(sums[i]))
    return out, sums


# ---- host functions around the hot path (oracle/prep_oracle.c) -------------------------------------------------------------
class Attr(C.Structure):
    """orc_attr = svr_image_attr, field for field"""
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double),
                ("xaxis", C.c_double * 3), ("yaxis", C.c_double * 3), ("zaxis", C.c_double * 3), ("origin", C.c_double * 3)]

    @classmethod
    def of(cls, a):
        return cls(int(a.nx), int(a.ny), int(a.nz), float(a.dx), float(a.dy), float(a.dz),
                   (C.c_double * 3)(*[float(v) for v in a.xaxis]), (C.c_double * 3)(*[float(v) for v in a.yaxis]),
                   (C.c_double * 3)(*[float(v) for v in a.zaxis]), (C.c_double * 3)(*[float(v) for v in a.origin]))


def match_stack_intensities(stacks, attrs, transformations, mask, mask_attr, average_value, together=False):
    """orc_match_stack_intensities (MatchStackIntensitiesWithMasking, RG.cc:1375-1493).  stacks: list of float64 [nz][ny][nx]
    (copied); returns (rescaled stacks, factors float32, per-stack averages)."""
    n = len(stacks)
    data = [np.ascontiguousarray(s, np.float64).copy() for s in stacks]
    at = (Attr * n)(*[Attr.of(a) for a in attrs])
    ptrs = (C.c_void_p * n)(*[d.ctypes.data for d in data])
    T = np.ascontiguousarray(np.stack([np.asarray(t, np.float64).reshape(16) for t in transformations]))
    m = np.ascontiguousarray(mask, np.float64)
    fac = np.zeros(n, np.float32)
    avg = np.zeros(n, np.float64)
    lib().orc_match_stack_intensities.restype = C.c_int
    rc = lib().orc_match_stack_intensities(n, at, ptrs, _p(T), C.byref(Attr.of(mask_attr)), _p(m), C.c_double(average_value), int(bool(together)),
                                           _p(fac), _p(avg))
    if rc:
        raise ValueError(f"stack {rc - 1} has no overlap with the ROI")
    return data, fac, avg


def generate_2d_patches(stack, attr, thickness, mask, mask_attr, pbbsize, stride, full_slices=False, snap=True, cap=None):
    """orc_generate_2d_patches (generate2DPatches, patchBasedObject.cuh:174-342): -> (patches float32 [n][py][px], i2w [n][16],
    w2i [n][16], total_pixels, origins [n][3])"""
    st = np.ascontiguousarray(stack, np.float32)
    mk = np.ascontiguousarray(mask, np.float32)
    px, py = (int(attr.nx), int(attr.ny)) if full_slices else (int(pbbsize[0]), int(pbbsize[1]))
    sx, sy = (px + 1, py + 1) if full_slices else (int(stride[0]), int(stride[1]))
    if cap is None:
        cap = attr.nz * len(range(0, attr.ny + py, sy)) * len(range(0, attr.nx + px, sx))
    data = np.zeros((cap, py, px), np.float32)
    i2w = np.zeros((cap, 16), np.float32)
    w2i = np.zeros((cap, 16), np.float32)
    org = np.zeros((cap, 3), np.float64)
    n = C.c_int(0)
    total = C.c_long(0)
    lib().orc_generate_2d_patches.restype = C.c_int
    rc = lib().orc_generate_2d_patches(C.byref(Attr.of(attr)), _p(st), C.c_double(thickness), C.byref(Attr.of(mask_attr)), _p(mk), px, py, sx, sy,
                                       int(bool(full_slices)), int(bool(snap)), int(cap), _p(data), _p(i2w), _p(w2i), _p(org), C.byref(n), C.byref(total))
    if rc:
        raise ValueError("more patches than the capacity")
    k = n.value
    return data[:k].copy(), i2w[:k].copy(), w2i[:k].copy(), int(total.value), org[:k].copy()


def segment_slic(stack, spx_size):
    """orc_segment_slic (runStackSLIC<T>::segmentSLIC, runStackSLIC.cpp:665-840): stack [nz][ny][nx] -> labels float32 [nz][ny][nx]"""
    st = np.ascontiguousarray(stack, np.float32)
    nz, ny, nx = st.shape
    out = np.zeros(st.shape, np.float32)
    lib().orc_segment_slic.restype = C.c_int
    if lib().orc_segment_slic(_p(st), nx, ny, nz, int(spx_size[0]), int(spx_size[1]), _p(out)):
        raise ValueError("superpixels larger than the slice")
    return out


def generate_2d_superpixel_patches(stack, attr, labels, thickness, mask, mask_attr, spx_size, extend_percent, cap=None):
    """orc_generate_2d_superpixel_patches (generate2DSuperpixelPatches, patchBasedObject.cuh:433-802) -> (patches float32 [n][py][px],
    spxMask uint8 [n][4096], i2w [n][16], w2i [n][16], origins [n][3], total_pixels)"""
    st = np.ascontiguousarray(stack, np.float32)
    lb = np.ascontiguousarray(labels, np.float32)
    mk = np.ascontiguousarray(mask, np.float32)
    px, py = min(64, int(attr.nx)), min(64, int(attr.ny))
    if cap is None:
        cap = int(attr.nz) * (int(lb.max()) + 2)
    data = np.zeros((cap, py, px), np.float32)
    sm = np.zeros((cap, 4096), np.uint8)
    i2w, w2i = np.zeros((cap, 16), np.float32), np.zeros((cap, 16), np.float32)
    org = np.zeros((cap, 3), np.float64)
    n, total = C.c_int(0), C.c_long(0)
    pxy = (C.c_int * 2)()
    lib().orc_generate_2d_superpixel_patches.restype = C.c_int
    rc = lib().orc_generate_2d_superpixel_patches(C.byref(Attr.of(attr)), _p(st), _p(lb), C.c_double(thickness), C.byref(Attr.of(mask_attr)), _p(mk),
                                                  int(spx_size[0]), int(spx_size[1]), int(extend_percent), int(cap), _p(data), _p(sm), _p(i2w),
                                                  _p(w2i), _p(org), C.byref(n), C.byref(total), pxy)
    if rc:
        raise ValueError("more patches than the capacity")
    assert (pxy[0], pxy[1]) == (px, py)
    k = n.value
    return data[:k].copy(), sm[:k].copy(), i2w[:k].copy(), w2i[:k].copy(), org[:k].copy(), int(total.value)


# ---- the rest of the pre-processing chain (round 3: CreateTemplate, SetMask, TransformMask, CropImage, MaskSlices) ---------
def _attr_to_py(a):
    from fetalreconstruction_amd import geometry as geo
    return geo.ImageAttributes(a.nx, a.ny, a.nz, a.dx, a.dy, a.dz, np.array(a.xaxis[:]), np.array(a.yaxis[:]), np.array(a.zaxis[:]), origin=np.array(a.origin[:]))


def split_packages(stack_attr, packages, evenodd=False, half=False, half_iter=1, max_packs=256, max_nz=1024):
    """orc_split_packages (SplitImage / SplitImageEvenOdd / SplitImageEvenOddHalf / HalfImage and PackageToVolume's slice assignment,
    RG.cc:4980-5192) -> list of (attributes, slices assigned from the geometry, stack planes held by construction)"""
    nz = np.zeros(max_packs, np.int32)
    sl = np.full((max_packs, max_nz), -1, np.int32)
    src = np.full((max_packs, max_nz), -1, np.int32)
    at = (Attr * max_packs)()
    lib().orc_split_packages.restype = C.c_int
    n = lib().orc_split_packages(C.byref(Attr.of(stack_attr)), int(packages), int(bool(evenodd)), int(bool(half)), int(half_iter), int(max_packs),
                                 int(max_nz), _p(nz), _p(sl), _p(src), at)
    if n < 0:
        raise ValueError("orc_split_packages: too many packages / planes")
    return [(_attr_to_py(at[p]), sl[p, :nz[p]].copy(), src[p, :nz[p]].copy()) for p in range(n)]


def create_template(stack_attr, resolution):
    """orc_create_template (CreateTemplate RG.cc:648-694) -> (template attributes, resolution used)"""
    out = Attr()
    lib().orc_create_template.restype = C.c_double
    d = lib().orc_create_template(C.byref(Attr.of(stack_attr)), C.c_double(resolution), C.byref(out))
    return _attr_to_py(out), float(d)


def gaussian_blur(data, attr, sigma):
    d = np.ascontiguousarray(data, np.float64).copy()
    lib().orc_gaussian_blur(C.byref(Attr.of(attr)), _p(d), C.c_double(sigma))
    return d


def set_mask(template_attr, mask, mask_attr, sigma, threshold=0.5):
    """orc_set_mask (SetMask RG.cc:750-803) -> (mask on the template grid [nz][ny][nx] float64, the blurred + binarised input)"""
    out = np.zeros((template_attr.nz, template_attr.ny, template_attr.nx), np.float64)
    if mask is None:
        lib().orc_set_mask(C.byref(Attr.of(template_attr)), None, None, C.c_double(sigma), C.c_double(threshold), _p(out))
        return out, None
    m = np.ascontiguousarray(mask, np.float64).copy()
    lib().orc_set_mask(C.byref(Attr.of(template_attr)), C.byref(Attr.of(mask_attr)), _p(m), C.c_double(sigma), C.c_double(threshold), _p(out))
    return out, m


def transform_mask(image, image_attr, mask, mask_attr, T):
    """orc_transform_mask (TransformMask RG.cc:805-821): the mask on the image's grid (the target starts as a copy of the image)"""
    tgt = np.ascontiguousarray(image, np.float64).copy()
    m = np.ascontiguousarray(mask, np.float64)
    t = np.ascontiguousarray(np.asarray(T, np.float64).reshape(16))
    lib().orc_transform_mask(C.byref(Attr.of(image_attr)), _p(tgt), C.byref(Attr.of(mask_attr)), _p(m), _p(t))
    return tgt


def crop_image(image, image_attr, mask):
    """orc_crop_bounds + GetRegion (CropImage RG.cc:5205-5306) -> (cropped image, its attributes, bounds6)"""
    b = (C.c_int * 6)()
    reg = Attr()
    m = np.ascontiguousarray(mask, np.float64)
    lib().orc_crop_bounds(C.byref(Attr.of(image_attr)), _p(m), b, C.byref(reg))
    x1, y1, z1, x2, y2, z2 = b[:]
    if x2 < x1 or y2 < y1 or z2 < z1:
        raise ValueError("CropImage: the mask does not overlap the image")
    return np.ascontiguousarray(np.asarray(image)[z1:z2 + 1, y1:y2 + 1, x1:x2 + 1]), _attr_to_py(reg), tuple(b[:])


def mask_slice(slice2d, slice_attr, T, mask, mask_attr):
    """orc_mask_slice (MaskSlices RG.cc:1940-1988) -> the masked slice [ny][nx] float64"""
    s = np.ascontiguousarray(slice2d, np.float64).copy()
    m = np.ascontiguousarray(mask, np.float64)
    t = np.ascontiguousarray(np.asarray(T, np.float64).reshape(16))
    lib().orc_mask_slice(C.byref(Attr.of(slice_attr)), _p(s), _p(t), C.byref(Attr.of(mask_attr)), _p(m))
    return s


def get_region_attr(attr, i1, j1, k1, i2, j2, k2):
    """attributes of GetRegion(i1, j1, k1, i2, j2, k2) (irtkGenericImage.cc:570-611)"""
    out = Attr()
    lib().orc_get_region_attr(C.byref(Attr.of(attr)), int(i1), int(j1), int(k1), int(i2), int(j2), int(k2), C.byref(out))
    return _attr_to_py(out)
