"""NIfTI-1 (.nii / .nii.gz) I/O through the C++ reader/writer of the engine library (csrc/svr_io.cpp),
with the reference's image conventions (irtkFileNIFTIToImage.cc:168-345, irtkImageToFileNIFTI.cc:65-145)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import engine as _engine
from . import geometry as geo
from .host import ImageAttr


def _lib():
    lib = _engine.load_library()
    lib.svr_free.restype = None
    return lib


def read(path):
    """-> (data float32 [nt][nz][ny][nx] squeezed to [nz][ny][nx] when nt == 1, geometry.ImageAttributes)"""
    lib = _lib()
    a, nt, ptr = ImageAttr(), C.c_int(0), C.POINTER(C.c_float)()
    err = C.create_string_buffer(256)
    rc = lib.svr_nifti_read(str(path).encode(), C.byref(a), C.byref(nt), C.byref(ptr), err)
    if rc != 0:
        raise _engine.SvrError(f"svr_nifti_read({path}): {err.value.decode()}")
    n = a.nx * a.ny * a.nz * nt.value
    data = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
    lib.svr_free(ptr)
    attr = geo.ImageAttributes(a.nx, a.ny, a.nz, a.dx, a.dy, a.dz, np.array(a.xaxis[:]), np.array(a.yaxis[:]),
                               np.array(a.zaxis[:]), np.array(a.origin[:]))
    data = data.reshape(nt.value, a.nz, a.ny, a.nx)
    return (data[0] if nt.value == 1 else data), attr


def write(path, data, attr):
    """data [nz][ny][nx] -> float32 NIfTI-1 with a qform (".gz" suffix = gzip)"""
    d = np.ascontiguousarray(data, np.float32)
    if d.shape != (attr.nz, attr.ny, attr.nx):
        raise _engine.SvrError(f"nifti.write: data {d.shape} does not match attributes {(attr.nz, attr.ny, attr.nx)}")
    err = C.create_string_buffer(256)
    a = ImageAttr.of(attr)
    rc = _lib().svr_nifti_write(str(path).encode(), C.byref(a), d.ctypes.data_as(C.c_void_p), err)
    if rc != 0:
        raise _engine.SvrError(f"svr_nifti_write({path}): {err.value.decode()}")


def read_dof(path):
    """IRTK rigid `dof` file -> (params [tx ty tz rx ry rz], 4x4 matrix)  (irtkRigidTransformation.cc:26-53, 392-426)"""
    p6, m = (C.c_double * 6)(), (C.c_double * 16)()
    err = C.create_string_buffer(256)
    if _lib().svr_dof_read(str(path).encode(), p6, m, err) != 0:
        raise _engine.SvrError(f"svr_dof_read({path}): {err.value.decode()}")
    return np.array(p6[:]), np.array(m[:]).reshape(4, 4)


def write_dof(path, params6):
    err = C.create_string_buffer(256)
    p6 = (C.c_double * 6)(*[float(v) for v in params6])
    if _lib().svr_dof_write(str(path).encode(), p6, err) != 0:
        raise _engine.SvrError(f"svr_dof_write({path}): {err.value.decode()}")
