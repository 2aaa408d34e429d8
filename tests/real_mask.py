"""The geometry and voxels of the reference's bundled brain mask (tests/golden/bundled_mask_bbox.npz, made by
tests/golden/make_mask_fixture.py) and synthetic stacks on that oblique, far-from-the-origin grid."""
import copy
import os

import numpy as np

from fetalreconstruction_amd import geometry as geo
from fetalreconstruction_amd import phantom

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bundled_mask_bbox.npz")
RADIUS = 60.0


def load():
    """-> (mask [z][y][x] float32 0/1, attributes of that crop, the npz)"""
    f = np.load(FIXTURE)
    shape = tuple(int(v) for v in f["crop_shape"])
    m = np.unpackbits(f["bits"])[: int(np.prod(shape))].reshape(shape).astype(np.float32)
    nz, ny, nx = shape
    a = geo.ImageAttributes(nx, ny, nz, *[float(v) for v in f["voxel"]], f["xaxis"].copy(), f["yaxis"].copy(), f["zaxis"].copy())
    a.origin = np.zeros(3)
    a.origin = f["first_voxel_world"] - (geo.image_to_world(a) @ np.array([0, 0, 0, 1.0]))[:3]
    return m, a, f


def centre(mask, attr):
    idx = np.argwhere(mask > 0).mean(0)
    return (geo.image_to_world(attr) @ np.array([idx[2], idx[1], idx[0], 1.0]))[:3]


def truth(attr, c):
    """the analytic phantom on a grid, scaled like phantom.make_stacks"""
    kk, jj, ii = np.meshgrid(np.arange(attr.nz), np.arange(attr.ny), np.arange(attr.nx), indexing="ij")
    w = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(np.float64) @ geo.image_to_world(attr).T
    return phantom.phantom_intensity(w[..., :3] - c, RADIUS) * 700.0 / 0.55


def stacks_on_mask_grid(mask, attr, n=3, spacing=2.5, seed=5, noise=5.0):
    """n stacks that cover the mask's box: the first on the mask's own axes, the others with the axes permuted cyclically
    (so every stack is oblique to the world axes and to the others); in-plane voxels of the mask, `spacing` between slices."""
    rng = np.random.default_rng(seed)
    c = centre(mask, attr)
    box_centre = (geo.image_to_world(attr) @ np.array([(attr.nx - 1) / 2, (attr.ny - 1) / 2, (attr.nz - 1) / 2, 1.0]))[:3]
    axes = [np.asarray(attr.xaxis, np.float64), np.asarray(attr.yaxis, np.float64), np.asarray(attr.zaxis, np.float64)]
    extent = [attr.nx * attr.dx, attr.ny * attr.dy, attr.nz * attr.dz]
    out = []
    for k in range(n):
        p = [(0 + k) % 3, (1 + k) % 3, (2 + k) % 3]
        a = geo.ImageAttributes(int(extent[p[0]] / attr.dx), int(extent[p[1]] / attr.dy), int(extent[p[2]] / spacing), attr.dx, attr.dy, spacing,
                                axes[p[0]].copy(), axes[p[1]].copy(), axes[p[2]].copy(), origin=box_centre + 0.31 * (k + 1) * axes[k % 3])
        val = truth(a, c) + rng.normal(0.0, noise, (a.nz, a.ny, a.nx))
        out.append((np.maximum(val, 0.0).astype(np.float32), a))
    return out, c
