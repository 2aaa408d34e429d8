"""ctypes binding of libsvr_hip.so: `Reconstruction`, a method-for-method mirror of the reference's
`class Reconstruction` (source/reconstructionGPU2/include/reconstruction_cuda2.cuh:92-341).

Method names, argument meaning and call order are the reference's, so the parity tests read
like irtkReconstruction's own call sequence (irtkReconstructionGPU.cc:249-401, 2695-3440).
Errors raise `SvrError` instead of the reference's print + exit(-err).

There is no CPU fallback: if the HIP library is missing or no GPU is visible, construction fails.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVR_HIP_LIB: tuning builds made by `build.py --variant` (tools/exp_*.py); the product is lib/libsvr_hip.so
LIB_PATH = os.environ.get("SVR_HIP_LIB") or os.path.join(_HERE, "lib", "libsvr_hip.so")

# enum svr_buffer / svr_timer (include/svr_hip.h)
BUF_RECONSTRUCTED, BUF_VOL_WEIGHTS, BUF_ADDON, BUF_CONFIDENCE_MAP, BUF_MASK = 0, 1, 2, 3, 4
BUF_BIAS_VOLUME, BUF_SMOOTH_MASK = 5, 6
BUF_SLICES, BUF_WEIGHTS, BUF_SIMSLICES, BUF_SIMWEIGHTS, BUF_PSF_SUMS, BUF_BIAS = 10, 11, 12, 13, 14, 15
BUF_SIMINSIDE, BUF_VOXEL_COUNT = 20, 21
T_BACKPROJECT, T_FORWARD, T_GAUSS, T_REGULARIZE, T_ESTEP, T_MSTEP, T_SCALE, T_REGISTER = range(8)
TIMER_NAMES = ("backproject", "forward", "gauss", "regularize", "estep", "mstep", "scale", "register", "allreduce", "exchange_host", "coeff_build", "reduce_scatter", "allgather",
               "backproject_table", "forward_table", "forward_store", "backproject_store")

EXPORTS = [
    "svr_create", "svr_destroy", "svr_last_error", "svr_set_flags", "svr_set_option", "svr_get_option", "svr_set_spx_masks", "svr_init_reconstruction_volume",
    "svr_set_mask", "svr_init_storage_volumes", "svr_fill_slices", "svr_set_slice_dims",
    "svr_set_slice_matrices", "svr_generate_psf_volume", "svr_update_scale_vector",
    "svr_update_slice_weights", "svr_update_reconstructed", "svr_sync_cpu", "svr_get_vol_weights",
    "svr_gaussian_reconstruction", "svr_simulate_slices", "svr_initialize_em_values",
    "svr_initialize_robust_statistics", "svr_estep", "svr_mstep", "svr_calculate_scale_vector",
    "svr_superresolution", "svr_mask_volume", "svr_scale_volume", "svr_restore_slice_intensities",
    "svr_debug_get", "svr_debug_set", "svr_debug_probe_pixel", "svr_device_ptr", "svr_volume_voxels", "svr_set_stream",
    "svr_gaussian_reconstruction_local", "svr_gaussian_reconstruction_finish",
    "svr_superresolution_backproject", "svr_superresolution_update", "svr_robust_statistics_sums",
    "svr_mstep_sums", "svr_scale_volume_sums", "svr_scale_volume_apply", "svr_timer_get",
    "svr_unit_counts", "svr_fallbacks", "svr_clock_probe", "svr_slice_em_setup", "svr_slice_em_set_state", "svr_mstep_estep_device", "svr_slice_em_run", "svr_slice_em_apply_weights", "svr_slice_em_fetch", "svr_slice_em_set_patch_form", "svr_cell_stats", "svr_pair_pack", "svr_pair_unpack", "svr_timer_reset", "svr_timer_enable", "svr_timer_begin", "svr_timer_end", "svr_timer_add", "svr_counters", "svr_get_stream", "svr_device", "svr_device_count", "svr_combine_weights", "svr_update_stack_sizes", "svr_ncc_set_targets", "svr_ncc_set_source",
    "svr_ncc_evaluate", "svr_ncc_alloc_targets", "svr_pyr_upload", "svr_pyr_level", "svr_correct_bias", "svr_normalise_bias", "svr_normalise_bias_local",
    "svr_normalise_bias_finish", "svr_init_reg_storage_volumes", "svr_fill_reg_slices",
    "svr_update_resampled_slices_i2w", "svr_prepare_slice_to_volume_reg", "svr_register_slices_to_volume",
    "svr_reg_set_schedule", "svr_reg_evaluate_costs", "svr_reg_counters", "svr_pvr_cc_patches", "svr_pvr_register_patches",
    "svr_slab_plan", "svr_slab_rs_pack", "svr_slab_update", "svr_slab_finish", "svr_stream_sync",
    "svr_get_scale_vector", "svr_adopt_scale_vector", "svr_get_slice_inside", "svr_mstep_estep", "svr_mstep_sums_fetch", "svr_mstep_partial", "svr_mstep_estep_ranks",
]


class SvrError(RuntimeError):
    pass


_lib = None


def load_library(path: str = LIB_PATH):
    """Load libsvr_hip.so (fails loudly if it was not built: run `python -m fetalreconstruction_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(path):
            raise SvrError(f"{path} not found: build the HIP engine first (fetalreconstruction_amd/build.py); "
                           "there is no CPU fallback")
        lib = C.CDLL(path)
        lib.svr_last_error.restype = C.c_char_p
        lib.svr_device_ptr.restype = C.c_void_p
        lib.svr_volume_voxels.restype = C.c_size_t
        lib.svr_destroy.restype = None
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u3(v):
    return (C.c_uint32 * 3)(*[int(x) for x in v])


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Reconstruction:
    """One GPU's SVR engine.  Mirrors `class Reconstruction` (RC.cuh:92-341)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.svr_create(int(device), C.byref(h))
        if rc != 0 or not h:
            raise SvrError(f"svr_create(device={device}) failed with status {rc} (no usable MI355X / HIP device?)")
        self._h = h
        self.vsize = None
        self.sgrid = None   # (ns, sy, sx)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.svr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            msg = self._lib.svr_last_error(self._h)
            raise SvrError(f"status {rc}: {msg.decode() if msg else ''}")

    # ---- state upload (names as in RC.cuh) --------------------------------------------
    def InitReconstructionVolume(self, size, dim, data=None, sigma_bias=12.0):
        d = _f32(data) if data is not None else None
        self._ck(self._lib.svr_init_reconstruction_volume(self._h, _u3(size), _f3(dim), _p(d) if d is not None else None,
                                                          C.c_float(sigma_bias)))
        self.vsize = tuple(int(s) for s in size)

    def setMask(self, size, dim, data, sigma_bias=12.0):
        d = _f32(data)
        self._ck(self._lib.svr_set_mask(self._h, _u3(size), _f3(dim), _p(d), C.c_float(sigma_bias)))

    def initStorageVolumes(self, size, dim):
        self._ck(self._lib.svr_init_storage_volumes(self._h, _u3(size), _f3(dim)))
        self.sgrid = (int(size[2]), int(size[1]), int(size[0]))

    def FillSlices(self, sdata, sizes_x, sizes_y):
        d = _f32(sdata)
        sx = np.ascontiguousarray(sizes_x, np.int32)
        sy = np.ascontiguousarray(sizes_y, np.int32)
        self._ck(self._lib.svr_fill_slices(self._h, _p(d), _p(sx), _p(sy)))

    def setSliceDims(self, slice_dims, quality_factor):
        d = _f32(slice_dims)
        self._ck(self._lib.svr_set_slice_dims(self._h, _p(d), C.c_float(quality_factor)))

    def SetSliceMatrices(self, slice_transforms, inv_slice_transforms, i2w_init, w2i_init, i2w, w2i, recon_i2w, recon_w2i):
        arrs = [_f32(a) for a in (slice_transforms, inv_slice_transforms, i2w_init, w2i_init, i2w, w2i, recon_i2w, recon_w2i)]
        self._ck(self._lib.svr_set_slice_matrices(self._h, *[_p(a) for a in arrs]))

    def generatePSFVolume(self, cpu_psf, psf_size, slice_voxel_dim, psf_dim, psf_i2w, psf_w2i, quality_factor):
        i2w, w2i = _f32(psf_i2w), _f32(psf_w2i)
        self._ck(self._lib.svr_generate_psf_volume(self._h, None, _u3(psf_size), _f3(slice_voxel_dim), _f3(psf_dim),
                                                   _p(i2w), _p(w2i), C.c_float(quality_factor)))

    def UpdateScaleVector(self, scales, slice_weights):
        s, w = _f32(scales), _f32(slice_weights)
        self._ck(self._lib.svr_update_scale_vector(self._h, _p(s), _p(w)))

    def UpdateSliceWeights(self, slice_weights):
        w = _f32(slice_weights)
        self._ck(self._lib.svr_update_slice_weights(self._h, _p(w)))

    def UpdateReconstructed(self, size, data):
        d = _f32(data)
        self._ck(self._lib.svr_update_reconstructed(self._h, _u3(size), _p(d)))

    def syncCPU(self):
        out = np.empty(int(np.prod(self.vsize)), np.float32)
        self._ck(self._lib.svr_sync_cpu(self._h, _p(out)))
        return out

    def combineWeights(self):
        out = np.zeros(int(np.prod(self.vsize)), np.float32)
        self._ck(self._lib.svr_combine_weights(self._h, _p(out)))
        return out

    def updateStackSizes(self, stack_sizes):
        a = np.ascontiguousarray(stack_sizes, np.uint32).reshape(-1, 3)
        self._ck(self._lib.svr_update_stack_sizes(self._h, _p(a), len(a)))

    def getVolWeights(self):
        out = np.empty(int(np.prod(self.vsize)), np.float32)
        self._ck(self._lib.svr_get_vol_weights(self._h, _p(out)))
        return out

    # ---- compute ------------------------------------------------------------------------
    def GaussianReconstruction(self):
        n = C.c_int(0)
        self._ck(self._lib.svr_gaussian_reconstruction(self._h, C.byref(n)))
        return [n.value]          # voxel_num: one entry per device (RC.cu:2331)

    def SimulateSlices(self):
        inside = np.zeros(self.sgrid[0], np.uint8)
        self._ck(self._lib.svr_simulate_slices(self._h, _p(inside)))
        return inside.astype(bool)

    def InitializeEMValues(self):
        self._ck(self._lib.svr_initialize_em_values(self._h))

    def InitializeRobustStatistics(self):
        s = C.c_float(0)
        self._ck(self._lib.svr_initialize_robust_statistics(self._h, C.byref(s)))
        return s.value

    def EStep(self, m, sigma, mix):
        pot = np.zeros(self.sgrid[0], np.float32)
        self._ck(self._lib.svr_estep(self._h, C.c_float(m), C.c_float(sigma), C.c_float(mix), _p(pot)))
        return pot

    def MStep(self, it, step, sigma, mix):
        s, mx, m = C.c_float(sigma), C.c_float(mix), C.c_float(0)
        self._ck(self._lib.svr_mstep(self._h, int(it), C.c_float(step), C.byref(s), C.byref(mx), C.byref(m)))
        return s.value, mx.value, m.value

    def CalculateScaleVector(self):
        sc = np.zeros(self.sgrid[0], np.float32)
        self._ck(self._lib.svr_calculate_scale_vector(self._h, _p(sc)))
        return sc

    # the deferred forms (the C++ host objects use them: one wait for the device per SR iteration instead of four)
    def SimulateSlicesDeferred(self):
        self._ck(self._lib.svr_simulate_slices(self._h, None))

    def GetSliceInside(self):
        inside = np.zeros(self.sgrid[0], np.uint8)
        self._ck(self._lib.svr_get_slice_inside(self._h, _p(inside)))
        return inside.astype(bool)

    def CalculateScaleVectorDeferred(self):
        self._ck(self._lib.svr_calculate_scale_vector(self._h, None))

    def GetScaleVector(self):
        sc = np.zeros(self.sgrid[0], np.float32)
        self._ck(self._lib.svr_get_scale_vector(self._h, _p(sc)))
        return sc

    def AdoptScaleVector(self):
        self._ck(self._lib.svr_adopt_scale_vector(self._h))

    def MStepEStep(self, it, step, sigma, mix, want_scale=False, want_inside=False):
        """MStep + EStep with one wait -> (sigma, mix, m, slice_potential, scale_vec | None, slice_inside | None)"""
        em = np.array([sigma, mix, 0.0], np.float32)
        pot = np.zeros(self.sgrid[0], np.float32)
        sc = np.zeros(self.sgrid[0], np.float32) if want_scale else None
        ins = np.zeros(self.sgrid[0], np.uint8) if want_inside else None
        self._ck(self._lib.svr_mstep_estep(self._h, int(it), C.c_float(step), _p(em), _p(pot), None if sc is None else _p(sc),
                                           None if ins is None else _p(ins)))
        return float(em[0]), float(em[1]), float(em[2]), pot, sc, None if ins is None else ins.astype(bool)

    def Superresolution(self, it, slice_weight, adaptive, alpha, min_intensity, max_intensity, delta, lam,
                        global_bias_correction=False, sigma_bias=12.0, low_intensity_cutoff=0.01):
        w = _f32(slice_weight)
        self._ck(self._lib.svr_superresolution(self._h, int(it), _p(w), int(bool(adaptive)), C.c_float(alpha),
                                               C.c_float(min_intensity), C.c_float(max_intensity), C.c_float(delta),
                                               C.c_float(lam), int(bool(global_bias_correction)), C.c_float(sigma_bias),
                                               C.c_float(low_intensity_cutoff)))

    def set_flags(self, disable_bias_correction=True, debug_gpu=False):
        self._ck(self._lib.svr_set_flags(self._h, int(bool(disable_bias_correction)), int(bool(debug_gpu))))

    def CorrectBias(self, sigma_bias, global_bias_correction=False):
        self._ck(self._lib.svr_correct_bias(self._h, C.c_float(sigma_bias), int(bool(global_bias_correction))))

    def NormaliseBias(self, it, sigma_bias):
        self._ck(self._lib.svr_normalise_bias(self._h, int(it), C.c_float(sigma_bias)))

    def maskVolume(self):
        self._ck(self._lib.svr_mask_volume(self._h))

    def ScaleVolume(self):
        self._ck(self._lib.svr_scale_volume(self._h))

    def RestoreSliceIntensities(self, stack_factors, stack_index):
        f = _f32(stack_factors)
        i = np.ascontiguousarray(stack_index, np.int32)
        self._ck(self._lib.svr_restore_slice_intensities(self._h, _p(f), len(f), _p(i)))

    # ---- sharded halves -----------------------------------------------------------------
    def GaussianReconstructionLocal(self):
        self._ck(self._lib.svr_gaussian_reconstruction_local(self._h))

    def GaussianReconstructionFinish(self):
        n = C.c_int(0)
        self._ck(self._lib.svr_gaussian_reconstruction_finish(self._h, C.byref(n)))
        return n.value

    def SuperresolutionBackproject(self, slice_weight=None):
        w = _f32(slice_weight) if slice_weight is not None else None
        self._ck(self._lib.svr_superresolution_backproject(self._h, _p(w) if w is not None else None))

    def SuperresolutionUpdate(self, adaptive, alpha, min_intensity, max_intensity, delta, lam):
        self._ck(self._lib.svr_superresolution_update(self._h, int(bool(adaptive)), C.c_float(alpha),
                                                      C.c_float(min_intensity), C.c_float(max_intensity),
                                                      C.c_float(delta), C.c_float(lam)))

    def RobustStatisticsSums(self):
        o = np.zeros(2, np.float64)
        self._ck(self._lib.svr_robust_statistics_sums(self._h, _p(o)))
        return o

    def MStepSums(self):
        o = np.zeros(5, np.float64)
        self._ck(self._lib.svr_mstep_sums(self._h, _p(o)))
        return o

    def MStepSumsFetch(self, want_scale=False, want_inside=False):
        """the M-step's five sums with the deferred vectors in the same wait (what a sharded host calls)"""
        o = np.zeros(5, np.float64)
        sc = np.zeros(self.sgrid[0], np.float32) if want_scale else None
        ins = np.zeros(self.sgrid[0], np.uint8) if want_inside else None
        self._ck(self._lib.svr_mstep_sums_fetch(self._h, _p(o), None if sc is None else _p(sc), None if ins is None else _p(ins)))
        return o, sc, None if ins is None else ins.astype(bool)

    def ScaleVolumeSums(self):
        o = np.zeros(2, np.float64)
        self._ck(self._lib.svr_scale_volume_sums(self._h, _p(o)))
        return o

    def ScaleVolumeApply(self, scale):
        self._ck(self._lib.svr_scale_volume_apply(self._h, C.c_float(scale)))

    def set_stream(self, stream_handle):
        self._ck(self._lib.svr_set_stream(self._h, C.c_void_p(stream_handle)))

    def device_ptr(self, which):
        return self._lib.svr_device_ptr(self._h, int(which))

    def stream_sync(self):
        """wait for everything queued on the engine's stream"""
        self._ck(self._lib.svr_stream_sync(self._h))

    # ---- debug getters (debugWeights/Simslices/... RC.cuh:192-207) ------------------------
    def debug_get(self, which):
        ns, sy, sx = self.sgrid if self.sgrid else (0, 0, 0)
        if which < 10:
            out = np.empty(int(np.prod(self.vsize)), np.float32)
        elif which < 20:
            out = np.empty((ns, sy, sx), np.float32)
        elif which == BUF_SIMINSIDE:
            out = np.empty((ns, sy, sx), np.uint8)
        else:
            out = np.empty((ns, sy, sx), np.int32)
        self._ck(self._lib.svr_debug_get(self._h, int(which), _p(out), C.c_size_t(out.nbytes)))
        return out

    def debug_set(self, which, arr):
        a = np.ascontiguousarray(arr)
        self._ck(self._lib.svr_debug_set(self._h, int(which), _p(a), C.c_size_t(a.nbytes)))

    # ---- slice-to-volume registration cost ------------------------------------------------
    def ncc_set_targets(self, targets):
        t = np.ascontiguousarray(targets, np.int16)
        n, ty, tx = t.shape
        self._ck(self._lib.svr_ncc_set_targets(self._h, n, tx, ty, _p(t)))

    def ncc_set_source(self, source=None):
        if source is None:
            self._ck(self._lib.svr_ncc_set_source(self._h, None, None))
        else:
            s = np.ascontiguousarray(source, np.int16)
            self._ck(self._lib.svr_ncc_set_source(self._h, _u3(s.shape[::-1]), _p(s)))

    def ncc_evaluate(self, target_index, matrices):
        idx = np.ascontiguousarray(target_index, np.int32)
        m = np.ascontiguousarray(matrices, np.float64).reshape(len(idx), 16)
        sums = np.zeros((len(idx), 6), np.int64)
        ncc = np.zeros(len(idx), np.float64)
        self._ck(self._lib.svr_ncc_evaluate(self._h, len(idx), _p(idx), _p(m), _p(sums), _p(ncc)))
        return ncc, sums

    # ---- GPU slice-to-volume registration (RC.cuh:326-338) ------------------------------------
    def initRegStorageVolumes(self, W, H, ns, dim=(1.0, 1.0, 1.0)):
        self._reg_grid = (int(ns), int(H), int(W))
        self._ck(self._lib.svr_init_reg_storage_volumes(self._h, _u3((W, H, ns)), _f3(dim)))

    def FillRegSlices(self, sdata, slices_resampled_i2w=None):
        d = _f32(sdata)
        if d.shape != self._reg_grid:
            raise SvrError(f"FillRegSlices: expected {self._reg_grid}, got {d.shape}")
        m = None if slices_resampled_i2w is None else _f32(slices_resampled_i2w).reshape(-1)
        self._ck(self._lib.svr_fill_reg_slices(self._h, _p(d), None if m is None else _p(m)))

    def updateResampledSlicesI2W(self, ofs):
        m = _f32(ofs).reshape(-1)
        if m.size != 16 * self._reg_grid[0]:
            raise SvrError("updateResampledSlicesI2W: one Matrix4 per slice expected")
        self._ck(self._lib.svr_update_resampled_slices_i2w(self._h, _p(m)))

    def prepareSliceToVolumeReg(self, volume=None):
        """Snapshots the engine's current reconstruction (`volume` is ignored; the oracle twin takes it)."""
        self._ck(self._lib.svr_prepare_slice_to_volume_reg(self._h))

    def set_schedule(self, levels=None, steps=None, iterations=None):
        self._ck(self._lib.svr_reg_set_schedule(self._h, int(levels or 0), int(steps or 0), int(iterations or 0)))

    def registerSlicesToVolume(self, transf):
        t = _f32(transf).reshape(-1).copy()
        if t.size != 16 * self._reg_grid[0]:
            raise SvrError("registerSlicesToVolume: one Matrix4 per slice expected")
        self._ck(self._lib.svr_register_slices_to_volume(self._h, _p(t)))
        return t.reshape(-1, 4, 4)

    def evaluate_costs(self, transf, level, active=None):
        ns, H, W = self._reg_grid
        t = _f32(transf).reshape(-1)
        act = None if active is None else np.ascontiguousarray(active, np.int32)
        a = ns if act is None else len(act)
        sim = np.zeros(ns, np.float32)
        dbg = np.zeros((3, a, H, W), np.float32)
        self._ck(self._lib.svr_reg_evaluate_costs(self._h, _p(t), int(level), None if act is None else _p(act), a,
                                                  _p(sim), _p(dbg)))
        return sim, dbg

    def cc_patches(self, ri2w, tmats, level, buffers=None):
        """computeCCpatch for every patch (patchBased2D3DRegistration_gpu2.cu:130-190)."""
        n = self.sgrid[0]
        r, t = _f32(ri2w).reshape(n, 16), _f32(tmats).reshape(n, 16)
        b = None if buffers is None else _f32(buffers)
        out = np.zeros(n, np.float32)
        sums = np.zeros((n, 6), np.float64)
        self._ck(self._lib.svr_pvr_cc_patches(self._h, None if b is None else _p(b), _p(r), _p(t), int(level), _p(out),
                                              _p(sums)))
        return out, sums

    def register_patches(self, ri2w, mo, invmo, transformations):
        """PatchBased2D3DRegistration_gpu2<T>::run (patchBased2D3DRegistration_gpu2.cu:450-566) -> (T [n][16], Tinv [n][16],
        counters {launches, evaluations, patches})."""
        n = self.sgrid[0]
        r, m, mi = (_f32(a).reshape(n, 16) for a in (ri2w, mo, invmo))
        t = _f32(transformations).reshape(n, 16).copy()
        ti = np.zeros((n, 16), np.float32)
        c = np.zeros(3, np.int64)
        self._ck(self._lib.svr_pvr_register_patches(self._h, _p(r), _p(m), _p(mi), _p(t), _p(ti), _p(c)))
        return t, ti, c

    def reg_counters(self):
        c = np.zeros(4, np.int64)
        self._ck(self._lib.svr_reg_counters(self._h, _p(c)))
        return c

    def set_spx_masks(self, masks):
        if masks is None:
            self._ck(self._lib.svr_set_spx_masks(self._h, None))
        else:
            m = np.ascontiguousarray(masks, np.uint8)
            self._ck(self._lib.svr_set_spx_masks(self._h, _p(m)))

    def set_option(self, name, value):
        self._ck(self._lib.svr_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int(0)
        self._ck(self._lib.svr_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def probe_pixel(self, sl, px, py):
        v = np.zeros(4096, np.float32)
        c = np.zeros(3, np.int32)
        self._ck(self._lib.svr_debug_probe_pixel(self._h, int(sl), int(px), int(py), _p(v), _p(c)))
        return v, c

    # ---- measurement --------------------------------------------------------------------
    def timer_enable(self, on=True):
        self._ck(self._lib.svr_timer_enable(self._h, int(bool(on))))

    def timer_reset(self):
        self._ck(self._lib.svr_timer_reset(self._h))

    def timers(self):
        out = {}
        for i, name in enumerate(TIMER_NAMES):
            ms, n = C.c_double(0), C.c_long(0)
            self._ck(self._lib.svr_timer_get(self._h, i, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def unit_counts(self):
        o = (C.c_uint64 * 3)()
        self._ck(self._lib.svr_unit_counts(self._h, o))
        return dict(pixels=int(o[0]), live_units=int(o[1]), dead_units=int(o[2]))

    def slab_chunks(self, world, rank):
        """svr_slab_plan: floats per rank of the slab update's reduce-scatter and all-gather messages"""
        a, b = C.c_size_t(0), C.c_size_t(0)
        self._ck(self._lib.svr_slab_plan(self._h, int(world), int(rank), C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def cell_stats(self):
        o = (C.c_uint64 * 8)()
        self._ck(self._lib.svr_cell_stats(self._h, o))
        return dict(items=int(o[0]), runs=int(o[1]), sorted_pixels=int(o[2]), staging_bytes=int(o[3]), gather_items=int(o[4]),
                    gather_runs=int(o[5]), gather_partial_bytes=int(o[6]), cell=(int(o[7]) >> 32, int(o[7]) & 0xFFFFFFFF))

    def fallbacks(self):
        """launches that left the cell path since the context was made (svr_fallbacks): zeros on every bench workload"""
        o = (C.c_uint64 * 4)()
        self._ck(self._lib.svr_fallbacks(self._h, o))
        return dict(scatter_to_atomics=int(o[0]), gather_to_tiles=int(o[1]), gauss1_to_tiles=int(o[2]), tiles_rerun=int(o[3]))

    def clock_probe(self, chain=1 << 20):
        """(ms, chain): the time of `chain` packed f32 fmas per lane (eight independent chains) on 4 wavefronts per SIMD of the whole chip
        (svr_clock_probe) -- a fixed number of shader cycles at the vector pipe's full issue rate: boxes compare by it"""
        ms = C.c_double()
        self._ck(self._lib.svr_clock_probe(self._h, int(chain), C.byref(ms)))
        return ms.value, int(chain)

    def counters(self):
        o = (C.c_uint64 * 8)()
        self._ck(self._lib.svr_counters(self._h, o))
        return dict(Vs=int(o[0]), active=int(o[1]), Va=int(o[2]), Nv=int(o[3]), slices=int(o[4]),
                    tiles=int(o[5]), fallback_tiles=int(o[6]), rerun8_tiles=int(o[7]))


def device_count():
    """HIP devices visible to this process (0 without a GPU)"""
    return int(load_library().svr_device_count())


def sync_gpu(rec: Reconstruction, prob, quality_factor: float = 2.0):
    """irtkReconstruction::SyncGPU + generatePSFVolume + UpdateGPUTranformationMatrices
    (irtkReconstructionGPU.cc:249-328, 1496-1610, 372-401): upload one problem to the engine."""
    from . import geometry as geo

    vs, vd = prob.vsize, prob.vdim
    rec.InitReconstructionVolume(vs, vd, None, 12.0)
    rec.setMask(vs, vd, prob.mask, 12.0)
    ns, sy, sx = prob.slices.shape
    rec.initStorageVolumes((sx, sy, ns), tuple(prob.slice_dim[0]))
    rec.FillSlices(prob.slices, prob.sizes_x, prob.sizes_y)
    rec.setSliceDims(prob.slice_dim, quality_factor)
    a = geo.ImageAttributes(128, 128, 128, *[float(d) for d in vd])   # PSF_SIZE 128, RC.cuh:56
    rec.generatePSFVolume(None, (128, 128, 128), tuple(prob.slice_dim[0]), vd,
                          geo.to_matrix4(geo.image_to_world(a)), geo.to_matrix4(geo.world_to_image(a)),
                          quality_factor)
    rec.SetSliceMatrices(prob.slice_t, prob.slice_tinv, prob.slice_i2w, prob.slice_w2i, prob.slice_i2w,
                         prob.slice_w2i, prob.recon_i2w, prob.recon_w2i)
