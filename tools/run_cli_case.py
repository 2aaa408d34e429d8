"""Dev tool: NIfTI stacks with inter-stack motion through the command line, with and without registration
(correlation of the result with the analytic phantom)."""
import pathlib
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from fetalreconstruction_amd import geometry as geo, nifti, phantom
from tests.twins import cli  # noqa: E402

tmp = pathlib.Path(tempfile.mkdtemp())
R = 26.0
stacks, mask, mattr, rattr, rmask = phantom.make_stacks(4, (64, 64, 26), 1.1, 2.2, None, 1.0, R, seed=7, stack_motion_mm=2.5, stack_motion_deg=4.0)
t0 = stacks[0].transformation
paths = []
for k, st in enumerate(stacks):
    nifti.write(tmp / f"s{k}.nii.gz", st.data, st.attr)
    paths.append(str(tmp / f"s{k}.nii.gz"))
# the mask lives in the template's space: the phantom's mask pulled back through the template's motion
nifti.write(tmp / "mask.nii.gz", rmask, rattr)
for extra in (["--no_registration"], [], ["--useGPUReg"]):
    out = tmp / "o.nii.gz"
    t = time.time()
    cli.main(["-o", str(out), "-i", *paths, "-m", str(tmp / "mask.nii.gz"), "--resolution", "1.0", "--iterations", "3", "--rec_iterations_first", "4",
              "--rec_iterations_last", "8", "--smooth_mask", "0", *extra])
    dt = time.time() - t
    vol, va = nifti.read(out)
    kk, jj, ii = np.meshgrid(np.arange(va.nz), np.arange(va.ny), np.arange(va.nx), indexing="ij")
    w = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(float) @ (t0 @ geo.image_to_world(va)).T     # template space -> anatomy
    truth = phantom.phantom_intensity(w[..., :3], R)
    inside = (np.sum(w[..., :3] ** 2, -1) < (R - 4) ** 2) & (vol > 0)
    print("RESULT", extra, "corr", round(float(np.corrcoef(vol[inside], truth[inside])[0, 1]), 4), "voxels", int(inside.sum()), f"{dt:.1f} s", flush=True)
