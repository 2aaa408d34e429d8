"""ctypes binding of the C++ host object `svr::irtkReconstruction` (include/svr_host.h,
csrc/svr_host.cpp): the same operator surface as reconstruction.irtkReconstruction, but the host
logic (slice-level EM, M-step, sharding glue) runs in C++ like the reference's."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import engine as _engine

_AR_PAIR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
_AR_HOST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)
_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int)
_DEV2 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)      # reduce_scatter_device / allgather_device

COMM_EXPORTS = ["svr_comm_unique_id", "svr_comm_create", "svr_comm_rebind", "svr_comm_collectives", "svr_comm_world", "svr_comm_allreduce_host",
                "svr_comm_last_error", "svr_comm_destroy", "svr_group_create", "svr_group_uses_rccl", "svr_group_join", "svr_group_destroy"]                                                  # csrc/svr_rccl.cpp
HOST_EXPORTS = [
    "svrh_create", "svrh_destroy", "svrh_last_error", "svrh_set_intensity_range", "svrh_set_intensity_matching", "svrh_set_smoothing_parameters",
    "svrh_set_force_excluded", "svrh_initialize_em_values_gpu", "svrh_gaussian_reconstruction_gpu",
    "svrh_simulate_slices_gpu", "svrh_initialize_robust_statistics_gpu", "svrh_estep_gpu", "svrh_scale_gpu",
    "svrh_superresolution_gpu", "svrh_mstep_gpu", "svrh_mask_volume_gpu", "svrh_scale_volume_gpu",
    "svrh_sr_iteration", "svrh_reconstruct_iteration", "svrh_get_state", "svrh_set_bias_correction", "svrh_bias_gpu",
    "svrh_normalise_bias_gpu", "svrh_prepare_registration_slices", "svrh_slice_to_volume_registration_gpu",
    "svrh_get_registration_slices", "svrh_force_collectives", "svrh_set_slab_update", "svrh_set_unit_order",
]
PVR_HOST_EXPORTS = ["pvrh_create", "pvrh_destroy", "pvrh_last_error", "pvrh_initialize_em_values", "pvrh_initialize_robust_statistics",
                    "pvrh_estep", "pvrh_mstep", "pvrh_scale", "pvrh_reconstruct_iteration", "pvrh_register_patches", "pvrh_get_state",
                    "pvrh_create_sharded", "pvrh_force_collectives", "pvrh_set_slab_update", "pvrh_sr_iteration", "pvrh_set_unit_order"]      # csrc/pvr_host.cpp
IRTK_EXPORTS = ["svrh_stack_registrations", "svrh_slice_to_volume_registration", "svrh_package_to_volume", "svrh_irtk_resample_with_padding",
                "svrh_irtk_blur_with_padding", "svrh_irtk_rigid_parameters"]                                  # csrc/irtk_reg.cpp
IO_EXPORTS = ["svr_nifti_read", "svr_nifti_write", "svr_free", "svr_dof_read", "svr_dof_write", "svr_host_threads"]      # csrc/svr_io.cpp, declared in svr_host.h


class ImageAttr(C.Structure):
    """struct svr_image_attr (include/svr_host.h)"""
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("dx", C.c_double), ("dy", C.c_double),
                ("dz", C.c_double), ("xaxis", C.c_double * 3), ("yaxis", C.c_double * 3), ("zaxis", C.c_double * 3),
                ("origin", C.c_double * 3)]

    @classmethod
    def of(cls, a):
        """from a geometry.ImageAttributes"""
        return cls(int(a.nx), int(a.ny), int(a.nz), float(a.dx), float(a.dy), float(a.dz),
                   (C.c_double * 3)(*[float(v) for v in a.xaxis]), (C.c_double * 3)(*[float(v) for v in a.yaxis]),
                   (C.c_double * 3)(*[float(v) for v in a.zaxis]), (C.c_double * 3)(*[float(v) for v in a.origin]))


class _Coll(C.Structure):
    _fields_ = [("struct_size", C.c_size_t),          # sizeof of the struct the launcher knows (include/svr_host.h)
                ("user", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("allreduce_volume_pair", _AR_PAIR),
                ("allreduce_host", _AR_HOST), ("allgather_slices", _AG),
                # device-buffer reduce-scatter / all-gather of the slab update: not supplied by the torch.distributed callbacks (NULL ->
                # all-reduce + replicated update); on_engine_stream 0: the C++ host synchronises the engine's stream before ar_pair
                ("reduce_scatter_device", _DEV2), ("allgather_device", _DEV2), ("on_engine_stream", C.c_int)]


class RcclComm:
    """One rank's RCCL communicator, made and driven by the C library (include/svr_host.h `svr_comm_*`, csrc/svr_rccl.cpp): the
    volume pair is all-reduced in place on the engine's stream, the small host vectors through a device scratch buffer.
    `unique_id` = 128 bytes from `RcclComm.unique_id()` on ONE rank, handed to all (the launcher's job); the constructor
    blocks until every rank of `world` has called it."""

    @staticmethod
    def unique_id():
        lib = _engine.load_library()
        buf = C.create_string_buffer(128)
        if lib.svr_comm_unique_id(buf):
            raise _engine.SvrError("svr_comm_unique_id failed (librccl not found?)")
        return buf.raw

    def __init__(self, rec, rank, world, unique_id):
        self._lib = _engine.load_library()
        self._lib.svr_comm_create.restype = C.c_void_p
        self._lib.svr_comm_collectives.restype = C.c_void_p
        self._lib.svr_comm_last_error.restype = C.c_char_p
        self._lib.svr_comm_destroy.restype = None
        self.rank, self.world = int(rank), int(world)
        assert len(unique_id) == 128
        h = self._lib.svr_comm_create(self.rank, self.world, C.c_char_p(unique_id), rec._h)
        if not h:
            raise _engine.SvrError("svr_comm_create failed (see stderr)")
        self._h = C.c_void_p(h)
        self.collectives = C.c_void_p(self._lib.svr_comm_collectives(self._h))
        self._rec = rec                                   # the engine must outlive the communicator's use of its stream

    def rebind(self, rec):
        """the same communicator for another engine context on the same device (bench.py: a second workload in the same launch)"""
        if self._lib.svr_comm_rebind(self._h, rec._h):
            raise _engine.SvrError("svr_comm_rebind: " + self._lib.svr_comm_last_error(self._h).decode())
        self._rec = rec

    def rccl_world(self):
        """the communicator's size as RCCL reports it (ncclCommCount)"""
        return int(self._lib.svr_comm_world(self._h))

    def _host(self, a, op):
        v = np.ascontiguousarray(a, np.float64).copy()
        if self._lib.svr_comm_allreduce_host(self._h, v.ctypes.data_as(C.c_void_p), int(v.size), int(op)):
            raise _engine.SvrError("svr_comm_allreduce_host: " + self._lib.svr_comm_last_error(self._h).decode())
        return v

    def allreduce_sum(self, a):
        return self._host(a, 0)

    def allreduce_min(self, a):
        return self._host(a, 1)

    def allreduce_max(self, a):
        return self._host(a, 2)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.svr_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _torch_collectives(comm, n_local):
    """svr_collectives out of a sharding.TorchComm (torch.distributed callbacks; gloo in the tests): -> (the struct, the thunks to keep alive).
    Shared by the two host objects."""
    views = {}
    counts = [int(round(x)) for x in comm.allreduce_sum(np.eye(comm.world)[comm.rank] * n_local)]

    def ar_pair(user, ptr, n):
        try:
            from .sharding import _device_view
            if (ptr, n) not in views:
                views[(ptr, n)] = _device_view(comm.torch, ptr, n, comm.device)
            comm.dist.all_reduce(views[(ptr, n)], op=comm.dist.ReduceOp.SUM)
            comm.torch.cuda.synchronize()
            return 0
        except Exception as ex:      # never let an exception cross the C boundary
            print("allreduce_volume_pair failed:", ex)
            return 1

    def ar_host(user, data, n, op):
        try:
            a = np.ctypeslib.as_array(data, shape=(n,))
            r = (comm.allreduce_sum, comm.allreduce_min, comm.allreduce_max)[op](a.copy())
            a[:] = r
            return 0
        except Exception as ex:
            print("allreduce_host failed:", ex)
            return 1

    def ag(user, local, n_local, out, n_global):
        try:
            loc = np.ctypeslib.as_array(local, shape=(n_local,)).copy() if n_local else np.zeros(0, np.float32)
            g = comm.allgather_slices(loc, counts)
            np.ctypeslib.as_array(out, shape=(n_global,))[:] = g
            return 0
        except Exception as ex:
            print("allgather_slices failed:", ex)
            return 1

    # the two device-buffer collectives of the slab update (csrc/svr_slab.inc), staged through the host and added IN RANK ORDER
    # (TorchComm(slabs=True): what two ranks that share one GPU can run -- RCCL refuses them -- and, like the in-process group of
    # the command lines, bit for bit the replicated update).  on_engine_stream 0: the C++ host synchronises the engine's stream first.
    def rs_dev(user, send, recv, n):
        try:
            from .sharding import _device_view
            W = comm.world
            mine = _device_view(comm.torch, send, W * n, comm.device or "cuda").cpu()
            outs = [comm.torch.zeros_like(mine) for _ in range(W)]
            comm.dist.all_gather(outs, mine)
            acc = outs[0][comm.rank * n:(comm.rank + 1) * n].clone()
            for r in range(1, W):
                acc += outs[r][comm.rank * n:(comm.rank + 1) * n]
            _device_view(comm.torch, recv, n, comm.device or "cuda").copy_(acc)
            comm.torch.cuda.synchronize()
            return 0
        except Exception as ex:
            print("reduce_scatter_device failed:", ex)
            return 1

    def ag_dev(user, send, recv, n):
        try:
            from .sharding import _device_view
            W = comm.world
            mine = _device_view(comm.torch, send, n, comm.device or "cuda").cpu()
            outs = [comm.torch.zeros_like(mine) for _ in range(W)]
            comm.dist.all_gather(outs, mine)
            _device_view(comm.torch, recv, W * n, comm.device or "cuda").copy_(comm.torch.cat(outs))
            comm.torch.cuda.synchronize()
            return 0
        except Exception as ex:
            print("allgather_device failed:", ex)
            return 1

    slab_cbs = (_DEV2(rs_dev), _DEV2(ag_dev)) if getattr(comm, "slabs", False) else (_DEV2(0), _DEV2(0))
    cbs = (_AR_PAIR(ar_pair), _AR_HOST(ar_host), _AG(ag)) + slab_cbs
    return _Coll(C.sizeof(_Coll), None, comm.rank, comm.world, *cbs, 0), cbs


class irtkReconstruction:
    """C++ host object over one engine; `comm` is an RcclComm (the C library's RCCL collectives), a reconstruction.TorchComm
    (torch.distributed callbacks: gloo in the CPU tests) or None."""

    def __init__(self, rec: "_engine.Reconstruction", n_slices_global, slice_range=None, comm=None,
                 max_intensity=1.0, min_intensity=0.0, force_collectives=False):
        """force_collectives: a world-1 run goes through the communicator's callbacks like a sharded one (tests, bench.py)"""
        self._lib = _engine.load_library()
        self._lib.svrh_create.restype = C.c_void_p
        self._lib.svrh_last_error.restype = C.c_char_p
        self._lib.svrh_destroy.restype = None
        self._lib.svrh_set_intensity_range.restype = None
        self._lib.svrh_set_smoothing_parameters.restype = None
        self._lib.svrh_set_force_excluded.restype = None
        self.reconstructionGPU = rec
        self.ns = int(n_slices_global)
        self.lo, self.hi = slice_range if slice_range is not None else (0, self.ns)
        self._coll = None
        coll_ptr = None
        if isinstance(comm, RcclComm):
            if comm.world > 1 or force_collectives:
                self._comm = comm                       # the C library's own collectives (csrc/svr_rccl.cpp)
                coll_ptr = comm.collectives
        elif comm is not None and comm.world > 1:
            self._comm = comm
            self._coll, self._cbs = _torch_collectives(comm, self.hi - self.lo)     # (keep the thunks alive)
        if self._coll is not None:
            coll_ptr = C.byref(self._coll)
        h = self._lib.svrh_create(rec._h, self.ns, int(self.lo), int(self.hi), coll_ptr)
        if not h:
            raise _engine.SvrError("svrh_create failed")
        self._h = C.c_void_p(h)
        self._lib.svrh_set_intensity_range(self._h, C.c_double(min_intensity), C.c_double(max_intensity))
        if force_collectives:
            self._lib.svrh_force_collectives.restype = None
            self._lib.svrh_force_collectives(self._h, 1)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.svrh_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise _engine.SvrError(f"host status {rc}: {self._lib.svrh_last_error(self._h).decode()}")

    def SetSmoothingParameters(self, delta, lam):
        self._lib.svrh_set_smoothing_parameters(self._h, C.c_double(delta), C.c_double(lam))

    def set_slab_update(self, on):
        """sharded runs: the volume update by z-slabs (default) or all-reduce + the update replicated on every rank"""
        self._lib.svrh_set_slab_update.restype = None
        self._lib.svrh_set_slab_update(self._h, int(bool(on)))

    def set_unit_order(self, order):
        """order[k] = the reference's index of slice k of this object's numbering (sharding.shard_units); None = the same numbering"""
        a = None if order is None else np.ascontiguousarray(order, np.int32)
        assert a is None or len(a) == self.ns
        self._ck(self._lib.svrh_set_unit_order(self._h, None if a is None else a.ctypes.data_as(C.c_void_p)))

    def SetIntensityMatching(self, on):
        self._lib.svrh_set_intensity_matching.restype = None
        self._lib.svrh_set_intensity_matching(self._h, int(bool(on)))

    def SetForceExcludedSlices(self, idx):
        a = np.ascontiguousarray(idx, np.int32)
        self._lib.svrh_set_force_excluded(self._h, a.ctypes.data_as(C.c_void_p), len(a))

    def set_bias_correction(self, enable, sigma_bias=12.0):
        self._ck(self._lib.svrh_set_bias_correction(self._h, int(bool(enable)), C.c_double(sigma_bias)))

    def BiasGPU(self):
        self._ck(self._lib.svrh_bias_gpu(self._h))

    def NormaliseBiasGPU(self, it):
        self._ck(self._lib.svrh_normalise_bias_gpu(self._h, int(it)))

    def InitializeEMValuesGPU(self):
        self._ck(self._lib.svrh_initialize_em_values_gpu(self._h))

    InitializeEMGPU = InitializeEMValuesGPU

    def GaussianReconstructionGPU(self):
        self._ck(self._lib.svrh_gaussian_reconstruction_gpu(self._h))

    def SimulateSlicesGPU(self):
        self._ck(self._lib.svrh_simulate_slices_gpu(self._h))

    def InitializeRobustStatisticsGPU(self):
        self._ck(self._lib.svrh_initialize_robust_statistics_gpu(self._h))

    def EStepGPU(self):
        self._ck(self._lib.svrh_estep_gpu(self._h))

    def ScaleGPU(self):
        self._ck(self._lib.svrh_scale_gpu(self._h))

    def SuperresolutionGPU(self, it):
        self._ck(self._lib.svrh_superresolution_gpu(self._h, int(it)))

    def MStepGPU(self, it):
        self._ck(self._lib.svrh_mstep_gpu(self._h, int(it)))

    def MaskVolumeGPU(self):
        self._ck(self._lib.svrh_mask_volume_gpu(self._h))

    def ScaleVolumeGPU(self):
        self._ck(self._lib.svrh_scale_volume_gpu(self._h))

    def PrepareRegistrationSlices(self, slices, slice_attrs, recon_voxel):
        """irtkReconstruction::PrepareRegistrationSlices (RG.cc:2104-2181) in the C++ host; returns the packed
        resampled slices [n][y][x] it handed to the engine."""
        sl = np.ascontiguousarray(slices, np.float32)
        n, sy, sx = sl.shape
        attrs = (ImageAttr * n)(*[ImageAttr.of(a) for a in slice_attrs])
        self._ck(self._lib.svrh_prepare_registration_slices(self._h, sl.ctypes.data_as(C.c_void_p), sx, sy, attrs,
                                                            C.c_double(recon_voxel)))
        size = (C.c_int * 3)()
        self._ck(self._lib.svrh_get_registration_slices(self._h, size, None))
        out = np.zeros((size[2], size[1], size[0]), np.float32)
        self._ck(self._lib.svrh_get_registration_slices(self._h, size, out.ctypes.data_as(C.c_void_p)))
        self.reconstructionGPU._reg_grid = (size[2], size[1], size[0])
        return out

    def SliceToVolumeRegistrationGPU(self, transformations):
        """irtkReconstruction::SliceToVolumeRegistrationGPU (RG.cc:2214-2290) in the C++ host."""
        t = np.ascontiguousarray(transformations, np.float64).reshape(-1, 16).copy()
        self._ck(self._lib.svrh_slice_to_volume_registration_gpu(self._h, t.ctypes.data_as(C.c_void_p)))
        return t.reshape(-1, 4, 4)

    def sr_iteration(self, i):
        self._ck(self._lib.svrh_sr_iteration(self._h, int(i)))

    def reconstruct_iteration(self, rec_iterations, on_sr_iteration=None):
        self._ck(self._lib.svrh_reconstruct_iteration(self._h, int(rec_iterations)))

    def state(self):
        sc, sw, pot = (np.zeros(self.ns, np.float32) for _ in range(3))
        ins = np.zeros(self.ns, np.uint8)
        s8 = np.zeros(8, np.float64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        self._ck(self._lib.svrh_get_state(self._h, p(sc), p(sw), p(pot), p(ins), p(s8)))
        names = ("sigma", "mix", "m", "mean_s", "mean_s2", "sigma_s", "sigma_s2", "mix_s")
        return dict(scale=sc, slice_weight=sw, slice_potential=pot, slice_inside=ins.astype(bool),
                    **{k: float(v) for k, v in zip(names, s8)})

    # attribute-style access used by shared tests
    @property
    def _scale_gpu(self):
        return self.state()["scale"]

    @property
    def _slice_weight_gpu(self):
        return self.state()["slice_weight"]


class irtkPatchBasedReconstruction:
    """The C++ patch-to-volume loop (csrc/pvr_host.cpp) over one engine with option pvr=1; same member names as
    the Python mirror pvr.irtkPatchBasedReconstruction."""

    def __init__(self, rec: "_engine.Reconstruction", patches_per_stack, min_intensity, max_intensity, patch_range=None, comm=None,
                 force_collectives=False):
        """patch_range = (lo, hi) + comm (an RcclComm, or a sharding.TorchComm at world > 1): the engine holds the patches [lo, hi) of the global numbering
        (pvrh_create_sharded); patches_per_stack stays the global count per stack."""
        self._lib = _engine.load_library()
        self._lib.pvrh_create.restype = C.c_void_p
        self._lib.pvrh_create_sharded.restype = C.c_void_p
        self._lib.pvrh_last_error.restype = C.c_char_p
        self._lib.pvrh_destroy.restype = None
        self.e = rec
        c = np.ascontiguousarray(patches_per_stack, np.int32)
        self.n = int(c.sum())
        self.lo, self.hi = patch_range if patch_range is not None else (0, self.n)
        self._comm = comm
        self._coll = None
        if comm is None or isinstance(comm, RcclComm):
            use = comm is not None and (comm.world > 1 or force_collectives)
            coll_ptr = comm.collectives if use else None
        else:                                            # a sharding.TorchComm: torch.distributed callbacks (gloo between processes in the tests)
            use = comm.world > 1
            if use:
                self._coll, self._cbs = _torch_collectives(comm, self.hi - self.lo)     # (keep the thunks alive)
            coll_ptr = C.byref(self._coll) if use else None
        h = self._lib.pvrh_create_sharded(rec._h, c.ctypes.data_as(C.c_void_p), len(c), C.c_float(min_intensity), C.c_float(max_intensity),
                                          int(self.lo), int(self.hi), coll_ptr)
        if not h:
            raise _engine.SvrError("pvrh_create_sharded failed")
        self._h = C.c_void_p(h)
        if use and force_collectives:
            self._lib.pvrh_force_collectives.restype = None
            self._lib.pvrh_force_collectives(self._h, 1)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.pvrh_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise _engine.SvrError(f"host status {rc}: {self._lib.pvrh_last_error(self._h).decode()}")

    def set_slab_update(self, on):
        """sharded runs: the volume update by z-slabs (default) or all-reduce + the update replicated on every rank"""
        self._lib.pvrh_set_slab_update.restype = None
        self._lib.pvrh_set_slab_update(self._h, int(bool(on)))

    def set_unit_order(self, order):
        """order[k] = the global (stack after stack) index of patch k of this object's numbering; None = the same numbering"""
        a = None if order is None else np.ascontiguousarray(order, np.int32)
        assert a is None or len(a) == self.n
        self._ck(self._lib.pvrh_set_unit_order(self._h, None if a is None else a.ctypes.data_as(C.c_void_p)))

    def initializeEMValues(self):
        self._ck(self._lib.pvrh_initialize_em_values(self._h))

    def InitializeRobustStatistics(self):
        self._ck(self._lib.pvrh_initialize_robust_statistics(self._h))

    def EStep(self):
        self._ck(self._lib.pvrh_estep(self._h))

    def MStep(self, it):
        self._ck(self._lib.pvrh_mstep(self._h, int(it)))

    def Scale(self):
        self._ck(self._lib.pvrh_scale(self._h))

    def reconstruct_iteration(self, rec_iterations):
        self._ck(self._lib.pvrh_reconstruct_iteration(self._h, int(rec_iterations)))

    def sr_iteration(self, i):
        self._ck(self._lib.pvrh_sr_iteration(self._h, int(i)))

    def state(self):
        sc, pw, pot = (np.zeros(self.n, np.float32) for _ in range(3))
        s8 = np.zeros(8, np.float64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        self._ck(self._lib.pvrh_get_state(self._h, p(sc), p(pw), p(pot), p(s8)))
        names = ("m_sigma_gpu", "m_mix_gpu", "m_m_gpu", "m_mean_s_gpu", "m_mean_s2_gpu", "m_sigma_s_gpu", "m_sigma_s2_gpu", "m_mix_s_gpu")
        return dict(scale=sc, patch_weight=pw, patch_potential=pot, **{k: float(v) for k, v in zip(names, s8)})


# ---- the IRTK registration schedule around the batched NCC cost (csrc/irtk_reg.cpp) -----------------------------------
_SET_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16))
_SET_S = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int16))
_EVAL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double))


class _NccBackend(C.Structure):
    _fields_ = [("user", C.c_void_p), ("set_targets", _SET_T), ("set_source", _SET_S), ("evaluate", _EVAL)]


class NccBackend:
    """struct svr_ncc_backend over a Python evaluator `fn(target int16 [ty][tx], M float64 [4][4], source int16 [z][y][x]) ->
    six integer moments`: how the CPU tests put the oracle's restatement of irtkImageRigidRegistrationWithPadding::Evaluate
    behind the C++ schedule.  The product path passes no backend (the engine evaluates)."""

    def __init__(self, fn):
        self.fn, self.targets, self.source, self.calls = fn, None, None, 0

        def set_targets(user, n, tx, ty, t):
            self.targets = np.ctypeslib.as_array(t, shape=(n, ty, tx)).copy()
            return 0

        def set_source(user, size, s):
            self.source = np.ctypeslib.as_array(s, shape=(size[2], size[1], size[0])).copy()
            return 0

        def evaluate(user, n, idx, mats, sums, ncc):
            try:
                i = np.ctypeslib.as_array(idx, shape=(n,))
                m = np.ctypeslib.as_array(mats, shape=(n, 4, 4))
                out = np.ctypeslib.as_array(sums, shape=(n, 6))
                for e in range(n):
                    out[e] = np.asarray(self.fn(self.targets[i[e]], m[e], self.source), np.int64)
                self.calls += 1
                return 0
            except Exception as ex:      # never let an exception cross the C boundary
                print("ncc backend failed:", ex)
                return 1

        self._cbs = (_SET_T(set_targets), _SET_S(set_source), _EVAL(evaluate))
        self.struct = _NccBackend(None, *self._cbs)


def _reg_lib():
    lib = _engine.load_library()
    lib.svrh_irtk_rigid_parameters.restype = None
    return lib


def StackRegistrations(rec, stacks, attrs, transformations, template_number, mask=None, mask_attr=None, backend=None, keep_origin=False):
    """irtkReconstruction::StackRegistrations (RG.cc:849-1001) -> (new transformations [n][4][4], number of evaluations).
    stacks[i]: float64 [nz][ny][nx]; `rec` is the engine (None only with a `backend`); keep_origin: the variant of the
    patch-based command line (irtkStack3D3DRegistration.cpp:164-228)."""
    lib = _reg_lib()
    n = len(stacks)
    data = [np.ascontiguousarray(s, np.float64) for s in stacks]
    ptrs = (C.c_void_p * n)(*[d.ctypes.data for d in data])
    at = (ImageAttr * n)(*[ImageAttr.of(a) for a in attrs])
    t = np.ascontiguousarray(transformations, np.float64).reshape(n, 16).copy()
    m = None if mask is None else np.ascontiguousarray(mask, np.float64)
    ma = None if mask is None else ImageAttr.of(mask_attr)
    nev, err = C.c_long(0), C.create_string_buffer(256)
    rc = lib.svrh_stack_registrations(rec._h if rec is not None else None, C.byref(backend.struct) if backend else None, n, at, ptrs,
                                      t.ctypes.data_as(C.c_void_p), int(template_number), C.byref(ma) if ma is not None else None,
                                      m.ctypes.data_as(C.c_void_p) if m is not None else None, 1 if keep_origin else 0, C.byref(nev), err)
    if rc != 0:
        raise _engine.SvrError(f"svrh_stack_registrations: {err.value.decode()}")
    return t.reshape(n, 4, 4), nev.value


def SliceToVolumeRegistration(rec, slices, slice_attrs, transformations, recon_attr, reconstructed, backend=None, no_resample=False):
    """irtkReconstruction::SliceToVolumeRegistration (RG.cc:1991-2059, 2291-2303), the reference's default registration
    -> (new transformations [n][4][4], number of evaluations).  slices: the padded float32 grid [n][sy][sx].
    no_resample: the patch-to-volume registration of the patch-based command line (patchBased2D3DRegistration.cpp:88-225)."""
    lib = _reg_lib()
    g = np.ascontiguousarray(slices, np.float32)
    n, sy, sx = g.shape
    at = (ImageAttr * n)(*[ImageAttr.of(a) for a in slice_attrs])
    t = np.ascontiguousarray(transformations, np.float64).reshape(n, 16).copy()
    vol = np.ascontiguousarray(reconstructed, np.float32)
    ra = ImageAttr.of(recon_attr)
    nev, err = C.c_long(0), C.create_string_buffer(256)
    rc = lib.svrh_slice_to_volume_registration(rec._h if rec is not None else None, C.byref(backend.struct) if backend else None, n,
                                               g.ctypes.data_as(C.c_void_p), sx, sy, at, t.ctypes.data_as(C.c_void_p), C.byref(ra),
                                               vol.ctypes.data_as(C.c_void_p), 1 if no_resample else 0, C.byref(nev), err)
    if rc != 0:
        raise _engine.SvrError(f"svrh_slice_to_volume_registration: {err.value.decode()}")
    return t.reshape(n, 4, 4), nev.value


def PackageToVolume(rec, stacks, attrs, pack_num, transformations, recon_attr, reconstructed, evenodd=False, half=False, half_iter=1,
                    backend=None):
    """irtkReconstruction::PackageToVolume (RG.cc:5096-5192) -> (new per-slice transformations [n][4][4], evaluations)."""
    lib = _reg_lib()
    n = len(stacks)
    data = [np.ascontiguousarray(s, np.float64) for s in stacks]
    ptrs = (C.c_void_p * n)(*[d.ctypes.data for d in data])
    at = (ImageAttr * n)(*[ImageAttr.of(a) for a in attrs])
    pk = (C.c_int * n)(*[int(p) for p in pack_num])
    t = np.ascontiguousarray(transformations, np.float64).reshape(-1, 16).copy()
    assert len(t) == sum(int(a.nz) for a in attrs), "one transformation per slice"
    vol = np.ascontiguousarray(reconstructed, np.float32)
    ra = ImageAttr.of(recon_attr)
    nev, err = C.c_long(0), C.create_string_buffer(256)
    rc = lib.svrh_package_to_volume(rec._h if rec is not None else None, C.byref(backend.struct) if backend else None, n, at, ptrs, pk,
                                    int(bool(evenodd)), int(bool(half)), int(half_iter), t.ctypes.data_as(C.c_void_p), C.byref(ra),
                                    vol.ctypes.data_as(C.c_void_p), C.byref(nev), err)
    if rc != 0:
        raise _engine.SvrError(f"svrh_package_to_volume: {err.value.decode()}")
    return t.reshape(-1, 4, 4), nev.value


def irtk_resample_with_padding(data, attr, size3, padding):
    lib = _reg_lib()
    d = np.ascontiguousarray(data, np.int16)
    oa = ImageAttr()
    lib.svrh_irtk_resample_with_padding(C.byref(ImageAttr.of(attr)), d.ctypes.data_as(C.c_void_p), C.c_double(size3[0]), C.c_double(size3[1]),
                                        C.c_double(size3[2]), int(padding), C.byref(oa), None, 0)
    out = np.zeros((oa.nz, oa.ny, oa.nx), np.int16)
    lib.svrh_irtk_resample_with_padding(C.byref(ImageAttr.of(attr)), d.ctypes.data_as(C.c_void_p), C.c_double(size3[0]), C.c_double(size3[1]),
                                        C.c_double(size3[2]), int(padding), C.byref(oa), out.ctypes.data_as(C.c_void_p), C.c_long(out.size))
    return out, oa


def irtk_blur_with_padding(data, attr, sigma, padding):
    d = np.ascontiguousarray(data, np.int16).copy()
    _reg_lib().svrh_irtk_blur_with_padding(C.byref(ImageAttr.of(attr)), d.ctypes.data_as(C.c_void_p), C.c_double(sigma), int(padding))
    return d


def irtk_rigid_parameters(matrix):
    m = np.ascontiguousarray(matrix, np.float64).reshape(16)
    p, r = np.zeros(6), np.zeros(16)
    _reg_lib().svrh_irtk_rigid_parameters(m.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
    return p, r.reshape(4, 4)
