"""P4-scale end-to-end run of bin/SVRreconstructionGPU (defaults of the reference: 4 iterations, 4 / 13 SR iterations, IRTK
registration): wall time of the whole command line on NIfTI files.  usage: run_cli_p4.py [extra CLI options]"""
import pathlib
import os, subprocess
import sys
import tempfile
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from fetalreconstruction_amd import build, geometry as geo, nifti, phantom  # noqa: E402

tmp = pathlib.Path(tempfile.mkdtemp())
R = 50.0
stacks, mask, mattr, rattr, rmask = phantom.make_stacks(4, (100, 93, 70), 1.17647, 1.25, 2.5, 1.0, R, seed=1, orientations=("ax", "cor", "sag", "ax"),
                                                        stack_motion_mm=2.0, stack_motion_deg=3.0)
paths = []
for k, st in enumerate(stacks):
    nifti.write(tmp / f"s{k}.nii.gz", st.data, st.attr)
    paths.append(str(tmp / f"s{k}.nii.gz"))
nifti.write(tmp / "mask.nii.gz", rmask, rattr)
os.environ.setdefault("SVR_CLI_TIMING", "1")      # the stages' wall times on stderr
t = time.time()
r = subprocess.run([build.CLI, "-o", str(tmp / "o.nii.gz"), "-i", *paths, "-m", str(tmp / "mask.nii.gz"), "--resolution", "1.0", *sys.argv[1:]],
                   capture_output=True, text=True)
dt = time.time() - t
print(r.stderr[-4000:])
print("exit", r.returncode, f"wall {dt:.2f} s")
vol, va = nifti.read(tmp / "o.nii.gz")
kk, jj, ii = np.meshgrid(np.arange(va.nz), np.arange(va.ny), np.arange(va.nx), indexing="ij")
w = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(float) @ (stacks[0].transformation @ geo.image_to_world(va)).T
inside = (np.sum(w[..., :3] ** 2, -1) < (R - 6) ** 2) & (vol > 0)
print("correlation with the phantom", round(float(np.corrcoef(vol[inside], phantom.phantom_intensity(w[..., :3], R)[inside])[0, 1]), 4), "voxels", int(inside.sum()))
