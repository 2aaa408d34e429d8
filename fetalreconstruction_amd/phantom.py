"""Seeded synthetic stacks: the bundled 3T stacks are not in the mount and
the GPU box has no data, so every config runs on an analytic phantom.

Produces exactly what irtkReconstruction::SyncGPU + UpdateGPUTranformationMatrices hand to the
engine (irtkReconstructionGPU.cc:249-401): a padded slice grid [ns][sy][sx] with -1 padding,
per-slice I2W/W2I/T/Tinv Matrix4 (float32), slice voxel dims, the volume grid + float mask.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import geometry as geo


@dataclass
class Stack:
    """one acquired stack as the patch-based hosts hold it (m_stacks / m_stack_transformations)"""
    data: np.ndarray                 # [nz][ny][nx]
    attr: geo.ImageAttributes
    transformation: np.ndarray       # 4x4 float64 (m_stack_transformations[i])
    thickness: float


@dataclass
class Problem:
    # volume
    vsize: tuple            # (vx, vy, vz)
    vdim: tuple             # voxel size (mm)
    recon_i2w: np.ndarray   # float32[16]
    recon_w2i: np.ndarray
    mask: np.ndarray        # float32 [vz][vy][vx]
    # slices
    slices: np.ndarray      # float32 [ns][sy][sx], -1 = padding / outside mask
    slice_i2w: np.ndarray   # float32 [ns][16]
    slice_w2i: np.ndarray
    slice_t: np.ndarray
    slice_tinv: np.ndarray
    slice_dim: np.ndarray   # float32 [ns][3]
    sizes_x: np.ndarray     # int32 [ns]
    sizes_y: np.ndarray
    stack_index: np.ndarray  # int32 [ns]
    psf_c0: np.ndarray      # float32[3]
    min_intensity: float
    max_intensity: float
    name: str = ""
    slice_attr: list = None  # geometry.ImageAttributes per slice (host-side registration prep)

    @property
    def ns(self):
        return self.slices.shape[0]

    @property
    def nvox(self):
        return int(np.prod(self.vsize))


_ELLIPSOIDS = [  # centre (fraction of radius), semi-axes (fraction), amplitude
    ((0.0, 0.0, 0.0), (0.85, 0.75, 0.8), 0.55),
    ((0.25, 0.1, 0.05), (0.3, 0.35, 0.3), 0.25),
    ((-0.3, -0.1, 0.1), (0.25, 0.3, 0.35), 0.2),
    ((0.05, 0.35, -0.2), (0.2, 0.15, 0.25), -0.15),
    ((-0.1, -0.35, -0.25), (0.18, 0.2, 0.15), 0.3),
    ((0.1, -0.05, 0.4), (0.15, 0.25, 0.12), 0.35),
]


def phantom_intensity(w: np.ndarray, radius: float) -> np.ndarray:
    """Analytic 3-D phantom: 6 ellipsoids + sinusoidal texture; w is [...,3] world mm."""
    x, y, z = (w[..., 0] / radius, w[..., 1] / radius, w[..., 2] / radius)
    v = np.zeros(x.shape, dtype=np.float64)
    for (cx, cy, cz), (ax, ay, az), amp in _ELLIPSOIDS:
        r2 = ((x - cx) / ax) ** 2 + ((y - cy) / ay) ** 2 + ((z - cz) / az) ** 2
        v += amp * (r2 < 1.0)
    v += 0.02 * np.sin(9.0 * x) * np.sin(7.0 * y + 0.3) * np.sin(8.0 * z + 0.7)
    return np.maximum(v, 0.0)


_ORIENT = {
    "ax": (np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0, 0, 1.0])),
    "cor": (np.array([1.0, 0, 0]), np.array([0, 0, 1.0]), np.array([0, -1.0, 0])),
    "sag": (np.array([0, 1.0, 0]), np.array([0, 0, 1.0]), np.array([1.0, 0, 0])),
}


def _rot_axes(axes, deg):
    r = geo.rigid_matrix(rz=deg)[:3, :3]
    return tuple(r @ a for a in axes)


def make_problem(
    n_stacks=4,
    stack_shape=(48, 48, 12),      # (nx, ny, n_slices)
    in_plane=1.0,
    spacing=2.5,
    thickness=None,
    recon_res=1.0,
    mask_radius=None,
    motion_frac=0.2,
    motion_mm=2.0,
    motion_deg=2.0,
    noise_sigma=5.0,
    average=700.0,
    seed=20260928,
    orientations=("ax", "cor", "sag", "ax30"),
    stack_offsets_mm=0.37,
    name="",
) -> Problem:
    """Build seeded stacks of the named shape around the world origin.

    `stack_offsets_mm` shifts each stack by an irrational-ish sub-voxel amount so that no slice
    pixel lands exactly on a voxel centre (the reference PSF is NaN there).
    """
    rng = np.random.default_rng(seed)
    nx, ny, nsl = stack_shape
    thickness = float(thickness if thickness is not None else spacing)
    fov = max(nx * in_plane, ny * in_plane, nsl * spacing)
    radius = float(mask_radius if mask_radius is not None else 0.4 * fov)

    # volume grid: isotropic, covers the mask radius plus a margin (CreateTemplate analogue,
    # irtkReconstructionGPU.cc:648-694)
    n_v = int(np.ceil((2.0 * radius + 6.0 * recon_res) / recon_res))
    vattr = geo.ImageAttributes(n_v, n_v, n_v, recon_res, recon_res, recon_res)
    r_i2w = geo.image_to_world(vattr)
    r_w2i = geo.world_to_image(vattr)
    c1 = ((np.arange(n_v) - (n_v - 1) / 2.0) * recon_res) ** 2   # axis-aligned isotropic grid
    mask = ((c1[:, None, None] + c1[None, :, None] + c1[None, None, :]) < radius * radius).astype(np.float32)

    ns = n_stacks * nsl
    slices = np.full((ns, ny, nx), -1.0, dtype=np.float32)
    m_i2w = np.zeros((ns, 16), np.float32)
    m_w2i = np.zeros((ns, 16), np.float32)
    m_t = np.zeros((ns, 16), np.float32)
    m_ti = np.zeros((ns, 16), np.float32)
    sdim = np.zeros((ns, 3), np.float32)
    stack_index = np.zeros(ns, np.int32)
    attrs = []

    py, px = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    pix = np.stack([px, py, np.zeros_like(px), np.ones_like(px)], -1).astype(np.float64)

    k = 0
    for st in range(n_stacks):
        o = orientations[st % len(orientations)]
        if o.startswith("ax") and len(o) > 2:
            axes = _rot_axes(_ORIENT["ax"], float(o[2:]))
        else:
            axes = _ORIENT[o]
        off = stack_offsets_mm * np.array([1.0 + 0.31 * st, 0.77 - 0.23 * st, 0.53 + 0.19 * st])
        sattr = geo.ImageAttributes(nx, ny, nsl, in_plane, in_plane, spacing, *axes, origin=off)
        s_i2w = geo.image_to_world(sattr)
        for j in range(nsl):
            centre = s_i2w @ np.array([(nx - 1) / 2.0, (ny - 1) / 2.0, float(j), 1.0])
            a = geo.ImageAttributes(nx, ny, 1, in_plane, in_plane, thickness, *axes, origin=centre[:3])
            i2w = geo.image_to_world(a)
            w2i = geo.world_to_image(a)
            if rng.random() < motion_frac:
                p = np.concatenate([rng.uniform(-motion_mm, motion_mm, 3), rng.uniform(-motion_deg, motion_deg, 3)])
            else:
                p = np.concatenate([rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.05, 0.05, 3)])
            t = geo.rigid_matrix(*p)
            w = (pix @ i2w.T) @ t.T
            val = phantom_intensity(w[..., :3], radius)
            inside = np.sum(w[..., :3] ** 2, -1) < radius * radius
            img = val * average / 0.55 + rng.normal(0.0, noise_sigma, val.shape)
            img = np.where(inside & (img >= 0.01), img, -1.0)  # MaskSlices, RG.cc:1956-1985
            slices[k] = img.astype(np.float32)
            m_i2w[k] = geo.to_matrix4(i2w)
            m_w2i[k] = geo.to_matrix4(w2i)
            m_t[k] = geo.to_matrix4(t)
            m_ti[k] = geo.to_matrix4(np.linalg.inv(t))
            sdim[k] = (in_plane, in_plane, thickness)
            stack_index[k] = st
            attrs.append(a)
            k += 1

    pos = slices[slices > 0]
    return Problem(
        vsize=(n_v, n_v, n_v),
        vdim=(recon_res,) * 3,
        recon_i2w=geo.to_matrix4(r_i2w),
        recon_w2i=geo.to_matrix4(r_w2i),
        mask=mask,
        slices=slices,
        slice_i2w=m_i2w,
        slice_w2i=m_w2i,
        slice_t=m_t,
        slice_tinv=m_ti,
        slice_dim=sdim,
        sizes_x=np.full(ns, nx, np.int32),
        sizes_y=np.full(ns, ny, np.int32),
        stack_index=stack_index,
        psf_c0=geo.psf_centre_offset((recon_res,) * 3),
        min_intensity=float(pos.min()) if pos.size else 0.0,   # InitializeEMGPU RG.cc:2937-2951
        max_intensity=float(pos.max()) if pos.size else 1.0,
        name=name,
        slice_attr=attrs,
    )


def make_stacks(n_stacks=3, stack_shape=(32, 32, 8), in_plane=1.1, spacing=2.2, thickness=None, recon_res=1.0,
                mask_radius=14.0, noise_sigma=5.0, average=700.0, seed=3, orientations=("ax", "cor", "sag"),
                stack_offsets_mm=0.37, stack_motion_mm=0.5, stack_motion_deg=1.0):
    """Whole stacks (3-D images + one rigid transformation each) for the patch-based path: what
    irtkPatchBasedReconstruction holds in m_stacks / m_stack_transformations / m_mask before patch
    extraction.  Returns (stacks, mask [z][y][x] uint8, mask_attr, recon_attr, recon_mask)."""
    rng = np.random.default_rng(seed)
    nx, ny, nsl = stack_shape
    thickness = float(thickness if thickness is not None else spacing)
    radius = float(mask_radius)
    n_v = int(np.ceil((2.0 * radius + 6.0 * recon_res) / recon_res))
    vattr = geo.ImageAttributes(n_v, n_v, n_v, recon_res, recon_res, recon_res)
    c1 = ((np.arange(n_v) - (n_v - 1) / 2.0) * recon_res) ** 2
    recon_mask = ((c1[:, None, None] + c1[None, :, None] + c1[None, None, :]) < radius * radius).astype(np.float32)
    stacks = []
    kk, jj, ii = np.meshgrid(np.arange(nsl), np.arange(ny), np.arange(nx), indexing="ij")
    pix = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(np.float64)
    for st in range(n_stacks):
        axes = _ORIENT[orientations[st % len(orientations)]]
        off = stack_offsets_mm * np.array([1.0 + 0.31 * st, 0.77 - 0.23 * st, 0.53 + 0.19 * st])
        sattr = geo.ImageAttributes(nx, ny, nsl, in_plane, in_plane, spacing, *axes, origin=off)
        p = np.concatenate([rng.uniform(-stack_motion_mm, stack_motion_mm, 3),
                            rng.uniform(-stack_motion_deg, stack_motion_deg, 3)])
        t = geo.rigid_matrix(*p)
        w = (pix @ geo.image_to_world(sattr).T) @ t.T
        val = phantom_intensity(w[..., :3], radius) * average / 0.55 + rng.normal(0.0, noise_sigma, w.shape[:-1])
        stacks.append(Stack(np.maximum(val, 0.0).astype(np.float32), sattr, t, thickness))
    return stacks, (recon_mask > 0).astype(np.uint8), vattr, vattr, recon_mask


def sub_problem(prob: Problem, lo: int, hi: int, select=None) -> Problem:
    """The slice shard [lo, hi) (or an explicit index list) of a problem, same volume."""
    import copy
    q = copy.copy(prob)
    idx = np.arange(lo, hi) if select is None else np.asarray(select)
    for name in ("slices", "slice_i2w", "slice_w2i", "slice_t", "slice_tinv", "slice_dim", "sizes_x", "sizes_y",
                 "stack_index"):
        setattr(q, name, np.ascontiguousarray(getattr(prob, name)[idx]))
    if prob.slice_attr is not None:
        q.slice_attr = [prob.slice_attr[int(i)] for i in idx]
    return q


# Named configurations (BASELINE.json `configs`)
def problem_tiny(seed=1):
    """Oracle-sized case: 3 stacks of 32x32x8, used by CPU/GPU parity tests.  The mask radius
    (14 voxels) exceeds the PSF support so interior pixels reach simweight > 0.99."""
    return make_problem(3, (32, 32, 8), 1.1, 2.2, None, 1.0, 14.0, seed=seed,
                        orientations=("ax", "cor", "sag"), name="tiny")


def problem_p4(seed=20260928):
    """P4s (the round-1 stand-in, axis-aligned; the bench's P4 is workloads.problem_p4): stands in for 'SVR on the bundled 4x3T stacks at 1.0 mm' (configs[0..1]).

    4 stacks of 100x93x70 on the bundled mask's native grid (1.17647 x 1.17647 x 1.25 mm voxels,
    thickness 2.5 = twice the z spacing, reconstruction.cc:422-431), ax/cor/sag/ax30, recon 1.0 mm.
    """
    return make_problem(4, (100, 93, 70), 1.17647, 1.25, 2.5, 1.0, 50.0, seed=seed, name="P4s")


def problem_s8(seed=20260928, n_stacks=8, slices_per_stack=64):
    """S8: 8 stacks x 64 slices x 256^2, 1.0 mm in-plane, 2.5 mm spacing/thickness, 0.75 mm recon."""
    return make_problem(n_stacks, (256, 256, slices_per_stack), 1.0, 2.5, 2.5, 0.75, 100.0, seed=seed,
                        orientations=("ax", "cor", "sag"), name="S8")
