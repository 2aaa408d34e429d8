"""Host-side geometry in the reference's (IRTK) conventions, float64 like the host code.

Mirrors:
  * image <-> world matrices: IRTKSimple2/image++/src/irtkBaseImage.cc:79-147
  * rigid 6-DOF parameters (mm, degrees) -> matrix:
    IRTKSimple2/packages/transformation/src/irtkRigidTransformation.cc:26-53
  * Matrix4 hand-over to the engine as row-major float32 (irtkReconstructionGPU.cc:330-342).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import copy

import numpy as np


@dataclass
class ImageAttributes:
    """irtkImageAttributes subset: size, voxel size, axes, origin (centre of the image)."""

    nx: int
    ny: int
    nz: int
    dx: float
    dy: float
    dz: float
    xaxis: np.ndarray = field(default_factory=lambda: np.array([1.0, 0.0, 0.0]))
    yaxis: np.ndarray = field(default_factory=lambda: np.array([0.0, 1.0, 0.0]))
    zaxis: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 1.0]))
    origin: np.ndarray = field(default_factory=lambda: np.zeros(3))


def mat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """4x4 product with irtkMatrix::operator*'s arithmetic (irtkMatrix.cc:222-242): every element a plain double sum from 0 over
    k = 0..3, no fused multiply-add, no BLAS reordering -- so the image <-> world matrices have the reference's last bits (a
    coordinate that is x.5 in exact arithmetic rounds to the same voxel as there; oracle/prep_oracle.c checks it)."""
    c = np.zeros((4, 4))
    for i in range(4):
        for j in range(4):
            t = 0.0
            for k in range(4):
                t += float(a[i, k]) * float(b[k, j])
            c[i, j] = t
    return c


def apply_points(m: np.ndarray, p: np.ndarray) -> np.ndarray:
    """irtkBaseImage::ImageToWorld / WorldToImage / irtkHomogeneousTransformation::Transform on an array of points [..., 3 or 4]:
    a = M00 x + M01 y + M02 z + M03 evaluated left to right per element (irtkBaseImage.h:425-468); returns [..., 3]."""
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    return np.stack([m[0, 0] * x + m[0, 1] * y + m[0, 2] * z + m[0, 3],
                     m[1, 0] * x + m[1, 1] * y + m[1, 2] * z + m[1, 3],
                     m[2, 0] * x + m[2, 1] * y + m[2, 2] * z + m[2, 3]], -1)


def image_to_world(a: ImageAttributes) -> np.ndarray:
    """irtkBaseImage::GetImageToWorldMatrix (irtkBaseImage.cc:79-111)."""
    t1 = np.eye(4)
    t1[0, 3] = -(a.nx - 1) / 2.0
    t1[1, 3] = -(a.ny - 1) / 2.0
    t1[2, 3] = -(a.nz - 1) / 2.0
    sc = np.diag([a.dx, a.dy, a.dz, 1.0])
    rot = np.eye(4)
    rot[:3, 0] = a.xaxis
    rot[:3, 1] = a.yaxis
    rot[:3, 2] = a.zaxis
    t2 = np.eye(4)
    t2[:3, 3] = a.origin
    return mat_mul(t2, mat_mul(rot, mat_mul(sc, t1)))


def region_origin(a: ImageAttributes, i1, j1, k1, region: ImageAttributes) -> np.ndarray:
    """Origin irtkGenericImage::GetRegion gives the sub-image starting at voxel (i1, j1, k1) (irtkGenericImage.cc:570-611):
    the world position of that voxel minus the position of voxel (0, 0, 0) of the region grid with its origin at zero.
    The arithmetic is kept literal: coordinates that land on x.5 after the later WorldToImage round the way the reference's do."""
    z = copy.copy(region)
    z.origin = np.zeros(3)
    p1 = apply_points(image_to_world(a), np.array([float(i1), float(j1), float(k1), 1.0]))
    p2 = apply_points(image_to_world(z), np.array([0.0, 0.0, 0.0, 1.0]))
    return (p1 - p2)[:3]


def world_to_image(a: ImageAttributes) -> np.ndarray:
    """irtkBaseImage::GetWorldToImageMatrix (irtkBaseImage.cc:113-147)."""
    t1 = np.eye(4)
    t1[:3, 3] = -np.asarray(a.origin)
    rot = np.eye(4)
    rot[0, :3] = a.xaxis
    rot[1, :3] = a.yaxis
    rot[2, :3] = a.zaxis
    sc = np.diag([1.0 / a.dx, 1.0 / a.dy, 1.0 / a.dz, 1.0])
    t2 = np.eye(4)
    t2[0, 3] = (a.nx - 1) / 2.0
    t2[1, 3] = (a.ny - 1) / 2.0
    t2[2, 3] = (a.nz - 1) / 2.0
    return mat_mul(t2, mat_mul(sc, mat_mul(rot, t1)))


def rigid_matrix(tx=0.0, ty=0.0, tz=0.0, rx=0.0, ry=0.0, rz=0.0) -> np.ndarray:
    """irtkRigidTransformation::UpdateMatrix (irtkRigidTransformation.cc:26-53); degrees."""
    crx, cry, crz = np.cos(np.deg2rad([rx, ry, rz]))
    srx, sry, srz = np.sin(np.deg2rad([rx, ry, rz]))
    m = np.eye(4)
    m[0, 0] = cry * crz
    m[0, 1] = cry * srz
    m[0, 2] = -sry
    m[0, 3] = tx
    m[1, 0] = srx * sry * crz - crx * srz
    m[1, 1] = srx * sry * srz + crx * crz
    m[1, 2] = srx * cry
    m[1, 3] = ty
    m[2, 0] = crx * sry * crz + srx * srz
    m[2, 1] = crx * sry * srz - srx * crz
    m[2, 2] = crx * cry
    m[2, 3] = tz
    return m


def to_matrix4(m: np.ndarray) -> np.ndarray:
    """double 4x4 -> row-major float32[16] (irtkReconstruction::toMatrix4)."""
    return np.ascontiguousarray(m, dtype=np.float64).astype(np.float32).reshape(16)


def psf_centre_offset(recon_dim, psf_size=128) -> np.ndarray:
    """The `d_PSFI2W * ((PSFsize-1)/2)` term of reconstruction_cuda2.cu:172 as float32[3].

    The host PSF image is origin-centred with the reconstruction's voxel size
    (irtkReconstructionGPU.cc:1534-1551), so the term is 0 up to float32 rounding of the
    float32 I2W matrix; it is evaluated here in float32 with the literal operation order of
    recon_volumeHelper.cuh:134-145 (the engine's svr_generate_psf_volume does the same).
    """
    a = ImageAttributes(psf_size, psf_size, psf_size, *[float(d) for d in recon_dim])
    m = to_matrix4(image_to_world(a)).reshape(4, 4)
    c = np.float32((psf_size - 1) * 0.5)
    v = np.array([c, c, c], dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    for k in range(3):
        acc = np.float32(m[k, 0] * v[0])
        acc = np.float32(acc + np.float32(m[k, 1] * v[1]))
        acc = np.float32(acc + np.float32(m[k, 2] * v[2]))
        acc = np.float32(acc + m[k, 3])
        out[k] = acc
    return out


def irtk_round(x: float) -> int:
    """round() of irtkCommon.h:85-88 (half away from zero)."""
    return int(x + 0.5) if x > 0 else int(x - 0.5)
