"""BASELINE.json configs[4] through the C++ command line on one GPU: 8 synthetic stacks of 64 slices of 256x256 (1 mm pixels,
2.5 mm spacing), a spherical mask of radius 100 mm, reconstruction at 0.5 mm (406^3 voxels), SLICO superpixel patches
(--spxSize 32 --spxExtend 2).  Writes the stacks as NIfTI, runs bin/PVRreconstructionGPU, reports wall time and the
correlation with the analytic phantom.  usage: run_cli_config5.py [superpixel|square] [recon mm]"""
import subprocess, sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np
from fetalreconstruction_amd import phantom, nifti, build, geometry as geo
mode = sys.argv[1] if len(sys.argv) > 1 else "superpixel"
res = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
R = 100.0
tmp = "/tmp/config5"; os.makedirs(tmp, exist_ok=True)
t0 = time.time()
stacks, mask, mattr, rattr, rmask = phantom.make_stacks(8, (256, 256, 64), 1.0, 2.5, 2.5, 1.0, R, seed=7, orientations=("ax", "cor", "sag"),
                                                        stack_motion_mm=0.0, stack_motion_deg=0.0)
paths = []
for k, st in enumerate(stacks):
    nifti.write(f"{tmp}/s{k}.nii.gz", st.data, st.attr); paths.append(f"{tmp}/s{k}.nii.gz")
nifti.write(f"{tmp}/mask.nii.gz", rmask.astype(np.float32), rattr)
print("stacks written in", round(time.time() - t0, 1), "s", flush=True)
opts = ["-s", "--spxSize", "32", "--spxExtend", "2"] if mode.startswith("superpixel") else ["--patchSize", "32", "32", "--patchStride", "16", "16"]
t0 = time.time()
os.environ["SVR_CLI_TIMING"] = "1"
if mode.startswith("svr"):          # configs[3] on one GPU: bin/SVRreconstructionGPU, `svr` without / `svrreg` with the registrations
    r = subprocess.run([build.CLI, "-o", f"{tmp}/out.nii.gz", "-i", *paths, "-m", f"{tmp}/mask.nii.gz", "--thickness", *["2.5"] * 8,
                        "--resolution", str(res), "--iterations", "2", "--rec_iterations_first", "3", "--rec_iterations_last", "5",
                        *([] if mode == "svrreg" else ["--no_registration"])], capture_output=True, text=True)
else:
    r = subprocess.run([build.PVR_CLI, "-o", f"{tmp}/out.nii.gz", "-i", *paths, "-m", f"{tmp}/mask.nii.gz", "--thickness", *["2.5"] * 8,
                        "--resolution", str(res), "--iterations", "1", "--sr_iterations", "3", *([] if mode.endswith("reg") else ["--no_registration"]), *opts],
                       capture_output=True, text=True)
print(mode, "command line rc", r.returncode, "wall", round(time.time() - t0, 1), "s")
print(r.stderr[-4000:])
if r.returncode == 0:
    vol, va = nifti.read(f"{tmp}/out.nii.gz")
    sub = (slice(None, None, 4),) * 3
    kk, jj, ii = np.meshgrid(np.arange(va.nz), np.arange(va.ny), np.arange(va.nx), indexing="ij")
    w = np.stack([ii[sub], jj[sub], kk[sub], np.ones_like(ii[sub])], -1).astype(float) @ geo.image_to_world(va).T
    truth = phantom.phantom_intensity(w[..., :3], R)
    v = vol[sub]
    inside = (np.sum(w[..., :3] ** 2, -1) < (R - 8) ** 2) & (v > 0)
    print("volume", vol.shape, "voxels > 0:", int((vol > 0).sum()), "correlation with the phantom:", round(float(np.corrcoef(v[inside], truth[inside])[0, 1]), 4))
