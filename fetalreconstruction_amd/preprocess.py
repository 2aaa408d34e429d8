"""The pre-processing chain between the image files and the engine, host side, float64
like the reference's host code.  Mirrors of irtkReconstruction methods (RG.cc =
source/reconstructionGPU2/irtkReconstructionGPU.cc):

  CreateTemplate                      RG.cc:648-694   (+ irtkResampling::Initialize, irtkResampling.cc:74-130)
  SetMask                             RG.cc:750-803   (+ irtkGaussianBlurring.cc:40-125, irtkConvolution_1D.cc:42-90)
  TransformMask                       RG.cc:805-821
  CropImage                           RG.cc:5205-5306
  MatchStackIntensitiesWithMasking    RG.cc:1375-1493
  CreateSlicesAndTransformations      RG.cc:1835-1880
  MaskSlices                          RG.cc:1940-1988
  SyncGPU's packing of the slices     RG.cc:249-328
Stack-to-stack registration (RG.cc:849-1001, IRTK's rigid registration) is not part of this module: the
caller supplies the stack transformations (the `dof` files of the reference's `-t` option, or identity).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass

import numpy as np

from . import geometry as geo
from .phantom import Problem
from .geometry import irtk_round


@dataclass
class Image:
    data: np.ndarray                  # [nz][ny][nx] float64
    attr: geo.ImageAttributes

    def copy(self):
        return Image(self.data.copy(), copy.copy(self.attr))


def _round_half_away(x):
    return np.where(x > 0, np.floor(x + 0.5), np.ceil(x - 0.5))      # irtkCommon.h:85-88


def _grid(a):
    kk, jj, ii = np.meshgrid(np.arange(a.nz), np.arange(a.ny), np.arange(a.nx), indexing="ij")
    return np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(np.float64)


def CreateTemplate(stack_attr, resolution):
    """RG.cc:648-694: the stack's grid, two slices taller, resampled (nearest neighbour, empty) to isotropic
    voxels: counts int(n * d_old / d_new), same axes and origin.  Returns (attributes, d)."""
    a = copy.copy(stack_attr)
    a.nz += 2
    if resolution <= 0:
        d = a.dx if (a.dx <= a.dy and a.dx <= a.dz) else (a.dy if a.dy <= a.dz else a.dz)
    else:
        d = float(resolution)
    n = [int(a.nx * a.dx / d), int(a.ny * a.dy / d), int(a.nz * a.dz / d)]
    size = [d, d, d]
    for k, old in enumerate((a.dx, a.dy, a.dz)):
        if n[k] < 1:
            n[k], size[k] = 1, old
    a.nx, a.ny, a.nz = n
    a.dx, a.dy, a.dz = size
    return a, d


def gaussian_blur(img: Image, sigma):
    """irtkGaussianBlurring<irtkRealPixel>(sigma).Run(): separable, kernel of 2 * round(4 sigma / d) + 1 taps
    per axis, normalised by the taps that fall inside the image (irtkConvolution_1D.cc:55-90)."""
    out = img.data.astype(np.float64).copy()
    a = img.attr
    for axis, d, n in ((2, a.dx, a.nx), (1, a.dy, a.ny), (0, a.dz, a.nz)):
        if axis == 0 and a.nz == 1:
            continue                                                   # GB.cc:91 skips z for single planes
        s = sigma / d
        half = irtk_round(4 * sigma / d)
        k = np.exp(-(np.arange(-half, half + 1) ** 2) / (2.0 * s * s))
        num = np.zeros_like(out)
        den = np.zeros(n)
        for t, w in zip(range(-half, half + 1), k):
            lo, hi = max(0, -t), min(n, n - t)
            if lo >= hi:
                continue
            dst = [slice(None)] * 3
            src = [slice(None)] * 3
            dst[axis], src[axis] = slice(lo, hi), slice(lo + t, hi + t)
            num[tuple(dst)] += w * out[tuple(src)]
            den[lo:hi] += w
        shape = [1, 1, 1]
        shape[axis] = n
        den = den.reshape(shape)
        out = np.where(den > 0, num / np.where(den > 0, den, 1.0), 0.0)
    return Image(out, copy.copy(a))


def transform_nn(source: Image, target_attr, transformation=None, source_padding=0.0):
    """irtkImageTransformation with the nearest-neighbour interpolator, target padding -1 on an all-zero
    target: every target voxel takes the source voxel nearest to T(world position), `source_padding`
    where that falls outside the source (RG.cc:782-793, 808-819)."""
    t = np.eye(4) if transformation is None else np.asarray(transformation, np.float64)
    # one matrix application at a time like the reference (ImageToWorld, Transform, WorldToImage): composing the matrices first
    # changes the last bits and flips coordinates that are x.5 in exact arithmetic (oracle/prep_oracle.c)
    q = geo.apply_points(geo.world_to_image(source.attr), geo.apply_points(t, geo.apply_points(geo.image_to_world(target_attr), _grid(target_attr))))
    idx = _round_half_away(q[..., :3]).astype(np.int64)
    sa = source.attr
    ok = ((idx[..., 0] >= 0) & (idx[..., 0] < sa.nx) & (idx[..., 1] >= 0) & (idx[..., 1] < sa.ny) &
          (idx[..., 2] >= 0) & (idx[..., 2] < sa.nz))
    v = source.data[np.clip(idx[..., 2], 0, sa.nz - 1), np.clip(idx[..., 1], 0, sa.ny - 1),
                    np.clip(idx[..., 0], 0, sa.nx - 1)]
    return Image(np.where(ok, v, source_padding).astype(np.float64), copy.copy(target_attr))


def SetMask(template_attr, mask: Image | None, sigma, threshold=0.5):
    """RG.cc:750-803: blur (sigma > 0) and re-binarise the mask, resample it onto the template grid."""
    if mask is None:
        return Image(np.ones((template_attr.nz, template_attr.ny, template_attr.nx)), copy.copy(template_attr))
    m = mask.copy()
    if sigma > 0:
        m = gaussian_blur(m, sigma)
        m.data = (m.data > threshold).astype(np.float64)
    return transform_nn(m, template_attr)


def TransformMask(image_attr, mask: Image, transformation):
    """RG.cc:805-821: the mask on the grid of `image` under the stack's transformation."""
    return transform_nn(mask, image_attr, transformation)


def get_region(img: Image, x1, y1, z1, x2, y2, z2):
    """irtkGenericImage::GetRegion(i1, j1, k1, i2, j2, k2): the sub-image keeps voxel positions."""
    a = copy.copy(img.attr)
    a.nx, a.ny, a.nz = x2 - x1, y2 - y1, z2 - z1
    a.origin = geo.region_origin(img.attr, x1, y1, z1, a)
    return Image(img.data[z1:z2, y1:y2, x1:x2].copy(), a)


def CropImage(image: Image, mask: Image):
    """RG.cc:5205-5306: bounding box of mask > 0 (the mask lives on the image's grid)."""
    nz = np.argwhere(mask.data > 0)
    if len(nz) == 0:
        # every bound runs off the end: x1 = n, x2 = -1 ... the reference would call GetRegion with an empty
        # range and throw; surface it
        raise ValueError("CropImage: mask does not overlap the image")
    (z1, y1, x1), (z2, y2, x2) = nz.min(0), nz.max(0)
    return get_region(image, int(x1), int(y1), int(z1), int(x2) + 1, int(y2) + 1, int(z2) + 1)


def MatchStackIntensitiesWithMasking(stacks, stack_transformations, mask: Image, average_value, together=False):
    """RG.cc:1375-1493.  Rescales the stacks in place (only voxels > 0) so that their mean inside the mask is
    `average_value`; returns the per-stack factors (`_stack_factor`, float)."""
    m_w2i = geo.world_to_image(mask.attr)
    ma = mask.attr
    averages = []
    for st, t in zip(stacks, stack_transformations):
        q = geo.apply_points(m_w2i, geo.apply_points(np.asarray(t, np.float64), geo.apply_points(geo.image_to_world(st.attr), _grid(st.attr))))   # RG.cc:1404-1410
        idx = _round_half_away(q[..., :3]).astype(np.int64)
        ok = ((idx[..., 0] >= 0) & (idx[..., 0] < ma.nx) & (idx[..., 1] >= 0) & (idx[..., 1] < ma.ny) &
              (idx[..., 2] >= 0) & (idx[..., 2] < ma.nz))
        mv = mask.data[np.clip(idx[..., 2], 0, ma.nz - 1), np.clip(idx[..., 1], 0, ma.ny - 1),
                       np.clip(idx[..., 0], 0, ma.nx - 1)]
        sel = ok & (mv == 1)
        if not sel.any():
            raise ValueError("MatchStackIntensitiesWithMasking: a stack has no overlap with the ROI")
        averages.append(float(st.data[sel].sum() / sel.sum()))
    glob = float(np.mean(averages))
    factors = []
    for st, av in zip(stacks, averages):
        f = average_value / (glob if together else av)
        factors.append(np.float32(f))
        st.data = np.where(st.data > 0, st.data * f, st.data)
    return np.array(factors, np.float32)


def CreateSlicesAndTransformations(stacks, stack_transformations, thickness):
    """RG.cc:1835-1880: one slice per z-plane of each stack, its z voxel size = the slice thickness, the
    stack's transformation.  Returns (slices [list of 2-D arrays], attrs, transformations, stack_index)."""
    slices, attrs, ts, ids = [], [], [], []
    for i, (st, t, th) in enumerate(zip(stacks, stack_transformations, thickness)):
        for j in range(st.attr.nz):
            r = get_region(st, 0, 0, j, st.attr.nx, st.attr.ny, j + 1)
            r.attr.dz = float(th)
            slices.append(r.data[0].copy())
            attrs.append(r.attr)
            ts.append(np.asarray(t, np.float64).copy())
            ids.append(i)
    return slices, attrs, ts, np.array(ids, np.int32)


def MaskSlices(slices, attrs, transformations, mask: Image):
    """RG.cc:1940-1988: values < 0.01 and pixels whose transformed position is outside the mask become -1."""
    m_w2i = geo.world_to_image(mask.attr)
    ma = mask.attr
    out = []
    for s, a, t in zip(slices, attrs, transformations):
        s = np.where(s < 0.01, -1.0, np.asarray(s, np.float64))
        q = geo.apply_points(m_w2i, geo.apply_points(np.asarray(t, np.float64), geo.apply_points(geo.image_to_world(a), _grid(a)[0])))   # RG.cc:1961-1972
        idx = _round_half_away(q[..., :3]).astype(np.int64)
        ok = ((idx[..., 0] >= 0) & (idx[..., 0] < ma.nx) & (idx[..., 1] >= 0) & (idx[..., 1] < ma.ny) &
              (idx[..., 2] >= 0) & (idx[..., 2] < ma.nz))
        mv = mask.data[np.clip(idx[..., 2], 0, ma.nz - 1), np.clip(idx[..., 1], 0, ma.ny - 1),
                       np.clip(idx[..., 0], 0, ma.nx - 1)]
        out.append(np.where(ok & (mv != 0), s, -1.0))
    return out


def build_problem(template_attr, mask: Image, slices, attrs, transformations, stack_index, name="files"):
    """irtkReconstruction::SyncGPU's view of the data (RG.cc:249-328): slices packed into the grid of the largest
    one, padded with -1; Matrix4 copies of the geometry; intensity range of InitializeEMGPU (RG.cc:2937-2951)."""
    ns = len(slices)
    mx, my = max(a.nx for a in attrs), max(a.ny for a in attrs)
    grid = np.full((ns, my, mx), -1.0, np.float32)
    for k, (s, a) in enumerate(zip(slices, attrs)):
        grid[k, :a.ny, :a.nx] = s
    pos = grid[grid > 0]
    rd = (template_attr.dx, template_attr.dy, template_attr.dz)
    return Problem(
        vsize=(template_attr.nx, template_attr.ny, template_attr.nz), vdim=rd,
        recon_i2w=geo.to_matrix4(geo.image_to_world(template_attr)),
        recon_w2i=geo.to_matrix4(geo.world_to_image(template_attr)),
        mask=np.ascontiguousarray(mask.data, np.float32), slices=grid,
        slice_i2w=np.stack([geo.to_matrix4(geo.image_to_world(a)) for a in attrs]),
        slice_w2i=np.stack([geo.to_matrix4(geo.world_to_image(a)) for a in attrs]),
        slice_t=np.stack([geo.to_matrix4(t) for t in transformations]),
        slice_tinv=np.stack([geo.to_matrix4(np.linalg.inv(t)) for t in transformations]),
        slice_dim=np.array([[a.dx, a.dy, a.dz] for a in attrs], np.float32),
        sizes_x=np.array([a.nx for a in attrs], np.int32), sizes_y=np.array([a.ny for a in attrs], np.int32),
        stack_index=np.asarray(stack_index, np.int32), psf_c0=geo.psf_centre_offset(rd),
        min_intensity=float(pos.min()) if pos.size else 0.0, max_intensity=float(pos.max()) if pos.size else 1.0,
        name=name, slice_attr=list(attrs))
