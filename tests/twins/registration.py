"""Host side of the GPU slice-to-volume registration (the reference's `--useGPUReg` path).

Mirrors, in float64 like the host code:
  * irtkResamplingWithPadding<VoxelType>::Initialize / Run
    (IRTKSimple2/image++/src/irtkResamplingWithPadding.cc:198-252, 254-443) -- trilinear
    resampling that ignores padded voxels,
  * irtkReconstruction::PrepareRegistrationSlices (irtkReconstructionGPU.cc:2104-2181),
  * irtkReconstruction::SliceToVolumeRegistrationGPU (irtkReconstructionGPU.cc:2214-2290).
The device work is behind `Reconstruction.{initRegStorageVolumes, FillRegSlices,
updateResampledSlicesI2W, prepareSliceToVolumeReg, registerSlicesToVolume}` (include/svr_hip.h).
"""
from __future__ import annotations

import copy

import numpy as np

from fetalreconstruction_amd import geometry as geo


from fetalreconstruction_amd.geometry import irtk_round  # noqa: E402,F401


def resample_with_padding(img: np.ndarray, attr: geo.ImageAttributes, new_size, pad=-1.0):
    """irtkResamplingWithPadding(new_x, new_y, new_z, pad).Run() on one image [nz][ny][nx].

    Returns (output [nz'][ny'][nx'] float64, output attributes).  A voxel is written when fewer than
    4 of its 8 neighbours are padding (out-of-bounds neighbours count as not padded) and the valid
    weights do not sum to 0 (RWP.cc:178-194); everything else is `pad`."""
    img = np.asarray(img, np.float64)
    nz, ny, nx = img.shape
    # Initialize (RWP.cc:214-252)
    new_n = [irtk_round(n * old / new) for n, old, new in
             zip((nx, ny, nz), (attr.dx, attr.dy, attr.dz), new_size)]
    new_d = list(new_size)
    for k, old in enumerate((attr.dx, attr.dy, attr.dz)):
        if new_n[k] < 1:
            new_n[k], new_d[k] = 1, old
    out_attr = copy.copy(attr)
    out_attr.nx, out_attr.ny, out_attr.nz = new_n
    out_attr.dx, out_attr.dy, out_attr.dz = new_d
    # Run (RWP.cc:279-437)
    kk, jj, ii = np.meshgrid(np.arange(new_n[2]), np.arange(new_n[1]), np.arange(new_n[0]), indexing="ij")
    p = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(np.float64)
    m = geo.world_to_image(attr) @ geo.image_to_world(out_attr)
    q = p @ m.T
    x, y, z = q[..., 0], q[..., 1], q[..., 2]
    u, v, w = np.floor(x).astype(int), np.floor(y).astype(int), np.floor(z).astype(int)
    dx, dy, dz = x - u, y - v, z - w
    val = np.zeros(x.shape)
    wsum = np.zeros(x.shape)
    npad = np.full(x.shape, 8, int)
    # the reference's order: (u,v,w) w1, (u,v,w+1) w2, (u,v+1,w) w3, (u,v+1,w+1) w4, (u+1,..) w5-w8
    for du, fx in ((0, 1 - dx), (1, dx)):
        for dv, fy in ((0, 1 - dy), (1, dy)):
            for dw, fz in ((0, 1 - dz), (1, dz)):
                wt = fx * fy * fz
                a, b, c = u + du, v + dv, w + dw
                inb = (a >= 0) & (a < nx) & (b >= 0) & (b < ny) & (c >= 0) & (c < nz)
                g = img[np.clip(c, 0, nz - 1), np.clip(b, 0, ny - 1), np.clip(a, 0, nx - 1)]
                ok = inb & (g != pad)
                val += np.where(ok, g * wt, 0.0)
                wsum += np.where(ok, wt, 0.0)
                npad -= (ok | ~inb).astype(int)
    good = (npad < 4) & (wsum > 0)
    out = np.where(good, val / np.where(good, wsum, 1.0), pad)
    return out, out_attr


class RegistrationSlices:
    """What PrepareRegistrationSlices leaves behind: `_slices_resampled` (attributes + plane 0 of
    the pixel data, packed into the padded grid the engine receives)."""

    def __init__(self, attrs, combined, i2w):
        self.attrs = attrs          # resampled image attributes, one per slice
        self.combined = combined    # float32 [ns][maxY][maxX], -1 padding
        self.i2w = i2w              # float32 [ns][16]


def PrepareRegistrationSlices(rec, slices, slice_attrs, recon_dx) -> RegistrationSlices:
    """irtkReconstruction::PrepareRegistrationSlices (RG.cc:2104-2181).

    slices[i]: 2-D array (ny, nx) with -1 padding, slice_attrs[i]: its image attributes (nz = 1)."""
    res, attrs, i2w = [], [], []
    for s, a in zip(slices, slice_attrs):
        sx, sy = int(a.nx), int(a.ny)
        t, ta = resample_with_padding(np.asarray(s, np.float64)[None, :sy, :sx], a, (recon_dx,) * 3, -1.0)
        res.append(t)
        attrs.append(ta)
        i2w.append(geo.to_matrix4(geo.image_to_world(ta)))
    ns = len(res)
    max_x = max(t.shape[2] for t in res)
    max_y = max(t.shape[1] for t in res)
    combined = np.full((ns, max_y, max_x), -1.0, np.float32)          # combinedStacks = -1 (RG.cc:2143)
    for i, t in enumerate(res):
        combined[i, :t.shape[1], :t.shape[2]] = t[0].astype(np.float32)   # plane 0 only (RG.cc:2152)
    a0 = attrs[0]
    rec.initRegStorageVolumes(max_x, max_y, ns, (a0.dx, a0.dy, a0.dz))
    rec.FillRegSlices(combined, np.stack(i2w))
    return RegistrationSlices(attrs, combined, np.stack(i2w))


def SliceToVolumeRegistrationGPU(rec, reg: RegistrationSlices, transformations, volume=None):
    """irtkReconstruction::SliceToVolumeRegistrationGPU (RG.cc:2214-2290).

    transformations: float64 [ns][4][4] (`_transformations_gpu`); returns the updated matrices."""
    mos, transf, ofs = [], [], []
    for a, m in zip(reg.attrs, np.asarray(transformations, np.float64).reshape(-1, 4, 4)):
        mo = np.eye(4)
        mo[:3, 3] = a.origin                         # offset.PutTranslation*(origin), RG.cc:2229-2236
        a0 = copy.copy(a)
        a0.origin = np.zeros(3)                      # PutOrigin(0,0,0), RG.cc:2228
        mos.append(mo)
        transf.append(geo.to_matrix4(m @ mo))
        ofs.append(geo.to_matrix4(geo.image_to_world(a0)))
    rec.updateResampledSlicesI2W(np.stack(ofs))
    rec.prepareSliceToVolumeReg(volume)
    out = rec.registerSlicesToVolume(np.stack(transf))
    res = []
    for mo, m in zip(mos, np.asarray(out, np.float64).reshape(-1, 4, 4)):
        res.append(m @ np.linalg.inv(mo))            # RG.cc:2262-2267
    return np.stack(res)
