"""Named bench / test workloads (BASELINE.json `configs`).

P4   -- SURVEY 8d's P4 on the REFERENCE'S geometry: the volume grid and the mask come from the one data file the
        reference bundles (its brain mask: an oblique acquisition 300-400 mm off the world origin), through the same
        pre-processing chain the command line runs (CreateTemplate -> SetMask -> CreateSlicesAndTransformations ->
        MaskSlices -> SyncGPU packing).  4 stacks on the mask's native grid cropped to its bounding box
        (100 x 93 x 70 voxels of 1.17647 x 1.17647 x 1.25 mm, thickness 2.5), axial / coronal / sagittal /
        axial rotated by 30 degrees in the mask's own frame, 1.0 mm reconstruction (117 x 109 x 90).
P4s  -- the round-1 stand-in: the same stacks around the world origin, axis-aligned, spherical mask (phantom.problem_p4).
S8   -- BASELINE configs[3]: 8 stacks of 64 x 256^2 slices, 0.75 mm (phantom.problem_s8).  S8h: the same at 0.5 mm.
"""
from __future__ import annotations

import os

import numpy as np

from . import geometry as geo
from . import phantom
from . import preprocess as pp

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK_FIXTURE = os.path.join(_ROOT, "tests", "golden", "bundled_mask_bbox.npz")
RADIUS = 60.0          # phantom radius (mm) that fills the bundled brain mask


def load_bundled_mask(path=MASK_FIXTURE):
    """-> (mask Image on the crop of the bundled mask's grid, world centre of its voxels)"""
    f = np.load(path)
    shape = tuple(int(v) for v in f["crop_shape"])
    m = np.unpackbits(f["bits"])[: int(np.prod(shape))].reshape(shape).astype(np.float64)
    nz, ny, nx = shape
    a = geo.ImageAttributes(nx, ny, nz, *[float(v) for v in f["voxel"]], f["xaxis"].copy(), f["yaxis"].copy(), f["zaxis"].copy())
    a.origin = np.zeros(3)
    a.origin = f["first_voxel_world"] - (geo.image_to_world(a) @ np.array([0, 0, 0, 1.0]))[:3]
    idx = np.argwhere(m > 0).mean(0)
    c = (geo.image_to_world(a) @ np.array([idx[2], idx[1], idx[0], 1.0]))[:3]
    return pp.Image(m, a), c


def _stack_axes(xa, ya, za, orient):
    if orient == "ax":
        return xa, ya, za
    if orient == "cor":
        return xa, za, -ya
    if orient == "sag":
        return ya, za, xa
    if orient.startswith("ax"):
        d = np.deg2rad(float(orient[2:]))
        return np.cos(d) * xa + np.sin(d) * ya, -np.sin(d) * xa + np.cos(d) * ya, za
    raise ValueError(orient)


def problem_p4(seed=20260928, resolution=1.0, smooth_mask=4.0, motion_frac=0.2, motion_mm=2.0, motion_deg=2.0,
               noise_sigma=5.0, average=700.0, stack_shape=(100, 93, 70), orientations=("ax", "cor", "sag", "ax30"),
               mask_path=MASK_FIXTURE, name="P4"):
    """SURVEY 8d P4 on the bundled mask's geometry; see the module docstring."""
    rng = np.random.default_rng(seed)
    mask, c = load_bundled_mask(mask_path)
    ma = mask.attr
    xa, ya, za = (np.asarray(v, np.float64) for v in (ma.xaxis, ma.yaxis, ma.zaxis))
    nzv = np.argwhere(mask.data > 0)
    lo, hi = nzv.min(0), nzv.max(0)
    box_centre = (geo.image_to_world(ma) @ np.array([(lo[2] + hi[2]) / 2.0, (lo[1] + hi[1]) / 2.0, (lo[0] + hi[0]) / 2.0, 1.0]))[:3]
    nx, ny, nsl = stack_shape
    dx, dy, dz = ma.dx, ma.dy, ma.dz
    thickness = 2.0 * dz                                     # reconstruction.cc:422-431
    stacks, attrs_st = [], []
    slice_imgs, slice_attrs, slice_ts, ids = [], [], [], []
    py, px = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    pix = np.stack([px, py, np.zeros_like(px), np.ones_like(px)], -1).astype(np.float64)
    for st, o in enumerate(orientations):
        axes = _stack_axes(xa, ya, za, o)
        off = 0.37 * np.array([1.0 + 0.31 * st, 0.77 - 0.23 * st, 0.53 + 0.19 * st])   # sub-voxel: no pixel on a voxel centre
        sattr = geo.ImageAttributes(nx, ny, nsl, dx, dy, dz, *[a.copy() for a in axes], origin=box_centre + off)
        attrs_st.append(sattr)
        s_i2w = geo.image_to_world(sattr)
        for j in range(nsl):
            centre = s_i2w @ np.array([(nx - 1) / 2.0, (ny - 1) / 2.0, float(j), 1.0])
            a = geo.ImageAttributes(nx, ny, 1, dx, dy, thickness, *[v.copy() for v in axes], origin=centre[:3])
            if rng.random() < motion_frac:
                p = np.concatenate([rng.uniform(-motion_mm, motion_mm, 3), rng.uniform(-motion_deg, motion_deg, 3)])
            else:
                p = np.concatenate([rng.uniform(-0.05, 0.05, 3), rng.uniform(-0.05, 0.05, 3)])
            # rigid motion about the slice centre (IRTK's dofs rotate about the world origin: 350 mm away a 2 degree
            # rotation would carry the slice 12 mm out of the mask)
            t0 = geo.rigid_matrix(*p)
            shift = np.eye(4)
            shift[:3, 3] = centre[:3]
            t = shift @ t0 @ np.linalg.inv(shift)
            w = (pix @ geo.image_to_world(a).T) @ t.T
            val = phantom.phantom_intensity(w[..., :3] - c, RADIUS)
            img = np.maximum(val * average / 0.55 + rng.normal(0.0, noise_sigma, val.shape), 0.0)
            slice_imgs.append(img)
            slice_attrs.append(a)
            slice_ts.append(t)
            ids.append(st)
    tattr, _ = pp.CreateTemplate(attrs_st[0], resolution)
    vmask = pp.SetMask(tattr, mask, smooth_mask)
    masked = pp.MaskSlices(slice_imgs, slice_attrs, slice_ts, vmask)
    return pp.build_problem(tattr, vmask, masked, slice_attrs, slice_ts, np.array(ids, np.int32), name=name)


def get(name, **kw):
    """workload by the name bench.py / the tests use"""
    if name == "P4":
        return problem_p4(**kw)
    if name == "P4s":
        return phantom.problem_p4(**kw)
    if name == "S8":
        return phantom.problem_s8(**kw)
    if name == "S8h":
        return phantom.make_problem(8, (256, 256, 64), 1.0, 2.5, 2.5, 0.5, 100.0, orientations=("ax", "cor", "sag"), name="S8h", **kw)
    raise ValueError(f"unknown workload {name}")
