"""Regenerates tests/golden/bundled_mask_bbox.npz: the one data file the reference bundles
(data/mask_10_3T_brain_smooth.nii.gz, the brain mask of its README example) reduced to what the tests need -- the image
geometry as this package's NIfTI reader reports it and the mask voxels of the bounding box (+2 voxels), bit-packed.
The geometry is the point: an oblique acquisition (no axis within 35 degrees of a world axis) hundreds of mm away from the
world origin, which the phantoms of phantom.py never produce.  Runs only where /root/reference exists:
    python tests/golden/make_mask_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from fetalreconstruction_amd import geometry as geo  # noqa: E402
from fetalreconstruction_amd import nifti  # noqa: E402

SRC = "/root/reference/data/mask_10_3T_brain_smooth.nii.gz"
MARGIN = 2


def main():
    d, a = nifti.read(SRC)
    nz = np.argwhere(d > 0)
    lo = np.maximum(nz.min(0) - MARGIN, 0)
    hi = np.minimum(nz.max(0) + 1 + MARGIN, d.shape)
    crop = d[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] > 0
    first = geo.image_to_world(a) @ np.array([lo[2], lo[1], lo[0], 1.0])          # world position of the crop's voxel (0,0,0)
    out = os.path.join(ROOT, "tests", "golden", "bundled_mask_bbox.npz")
    np.savez_compressed(out, full_shape=np.array(d.shape), voxel=np.array([a.dx, a.dy, a.dz]), origin=np.asarray(a.origin, np.float64),
                        xaxis=np.asarray(a.xaxis, np.float64), yaxis=np.asarray(a.yaxis, np.float64), zaxis=np.asarray(a.zaxis, np.float64),
                        lo=lo, hi=hi, first_voxel_world=first[:3], count=np.array(int((d > 0).sum())),
                        bits=np.packbits(crop.reshape(-1)), crop_shape=np.array(crop.shape))
    print(out, os.path.getsize(out), "bytes; bbox", hi - lo, "voxels set", int(crop.sum()))


if __name__ == "__main__":
    main()
