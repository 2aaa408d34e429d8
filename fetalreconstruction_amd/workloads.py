"""Named bench / test workloads (BASELINE.json `configs`).

P4   -- SURVEY 8d's P4 on the REFERENCE'S geometry: the volume grid and the mask come from the one data file the
        reference bundles (its brain mask: an oblique acquisition 300-400 mm off the world origin), through the same
        pre-processing chain the command line runs (CreateTemplate -> SetMask -> CreateSlicesAndTransformations ->
        MaskSlices -> SyncGPU packing).  4 stacks on the mask's native grid cropped to its bounding box
        (100 x 93 x 70 voxels of 1.17647 x 1.17647 x 1.25 mm, thickness 2.5), axial / coronal / sagittal /
        axial rotated by 30 degrees in the mask's own frame, 1.0 mm reconstruction (117 x 109 x 90).
P4s  -- the round-1 stand-in: the same stacks around the world origin, axis-aligned, spherical mask (phantom.problem_p4).
S8   -- BASELINE configs[3]: 8 stacks of 64 x 256^2 slices, 0.75 mm (phantom.problem_s8).  S8h: the same at 0.5 mm.
PVR4 -- BASELINE configs[2]: 32 x 32 patches with stride 16 of the P4 stacks (written as NIfTI files on the bundled mask's
        grid), 1.0 mm, cut by the product's own command line (bin/PVRreconstructionGPU --dumpProblem --dryRun: reading, mask
        transformation, intensity matching, patch extraction).
PVR8spx -- BASELINE configs[4]: SLICO superpixel patches (--spxSize 32 --spxExtend 2) of the 8 S8 stacks, 0.5 mm.
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


def load_pvr_dump(path, superpixel):
    """What `bin/PVRreconstructionGPU --dumpProblem <file> --dryRun` (csrc/pvr_cli.cpp) is about to hand to the engine, as a
    phantom.Problem the bindings can upload (engine.sync_gpu)."""
    raw = open(path, "rb").read()
    hdr = np.frombuffer(raw, np.int32, 8)
    ns, px, py, nst, vx, vy, vz, ver = [int(v) for v in hdr]
    assert ver == 1, "dump without the geometry block"
    o = 32
    counts = np.frombuffer(raw, np.int32, nst, o); o += 4 * nst
    vmin, vmax = np.frombuffer(raw, np.float32, 2, o); o += 8
    patches = np.frombuffer(raw, np.float32, ns * py * px, o).reshape(ns, py, px); o += 4 * ns * py * px
    i2w = np.frombuffer(raw, np.float32, ns * 16, o).reshape(ns, 16); o += 64 * ns
    mask = np.frombuffer(raw, np.float32, vx * vy * vz, o).reshape(vz, vy, vx); o += 4 * vx * vy * vz
    spx = None
    if superpixel:
        spx = np.frombuffer(raw, np.uint8, ns * 4096, o).reshape(ns, 4096); o += ns * 4096
    w2i = np.frombuffer(raw, np.float32, ns * 16, o).reshape(ns, 16); o += 64 * ns
    t = np.frombuffer(raw, np.float32, ns * 16, o).reshape(ns, 16); o += 64 * ns
    ti = np.frombuffer(raw, np.float32, ns * 16, o).reshape(ns, 16); o += 64 * ns
    dims = np.frombuffer(raw, np.float32, ns * 3, o).reshape(ns, 3); o += 12 * ns
    gi2w = np.frombuffer(raw, np.float32, 16, o); o += 64
    gw2i = np.frombuffer(raw, np.float32, 16, o); o += 64
    gdim = np.frombuffer(raw, np.float32, 3, o); o += 12
    assert o == len(raw)
    gd = tuple(float(v) for v in gdim)
    prob = phantom.Problem(vsize=(vx, vy, vz), vdim=gd, recon_i2w=gi2w.copy(), recon_w2i=gw2i.copy(), mask=mask.copy(),
                           slices=patches.copy(), slice_i2w=i2w.copy(), slice_w2i=w2i.copy(), slice_t=t.copy(), slice_tinv=ti.copy(),
                           slice_dim=dims.copy(), sizes_x=np.full(ns, px, np.int32), sizes_y=np.full(ns, py, np.int32),
                           stack_index=np.repeat(np.arange(nst, dtype=np.int32), counts), psf_c0=geo.psf_centre_offset(gd),
                           min_intensity=float(vmin), max_intensity=float(vmax), name="pvr-dump")
    prob.patches_per_stack = [int(c) for c in counts]
    prob.spx_masks = spx
    return prob


def p4_stacks(seed=20260928, noise_sigma=5.0, average=700.0, stack_shape=(100, 93, 70),
              orientations=("ax", "cor", "sag", "ax30")):
    """The four P4 stacks as volumes on the bundled mask's frame (no motion inside a stack: a patch-to-volume case starts
    from the stacks as acquired) -> [(data [z][y][x] float32, attributes)], mask Image"""
    rng = np.random.default_rng(seed)
    mask, c = load_bundled_mask()
    ma = mask.attr
    xa, ya, za = (np.asarray(v, np.float64) for v in (ma.xaxis, ma.yaxis, ma.zaxis))
    nzv = np.argwhere(mask.data > 0)
    lo, hi = nzv.min(0), nzv.max(0)
    box_centre = (geo.image_to_world(ma) @ np.array([(lo[2] + hi[2]) / 2.0, (lo[1] + hi[1]) / 2.0, (lo[0] + hi[0]) / 2.0, 1.0]))[:3]
    nx, ny, nsl = stack_shape
    out = []
    kk, jj, ii = np.meshgrid(np.arange(nsl), np.arange(ny), np.arange(nx), indexing="ij")
    vox = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(np.float64)
    for st, o in enumerate(orientations):
        axes = _stack_axes(xa, ya, za, o)
        off = 0.37 * np.array([1.0 + 0.31 * st, 0.77 - 0.23 * st, 0.53 + 0.19 * st])
        sattr = geo.ImageAttributes(nx, ny, nsl, ma.dx, ma.dy, ma.dz, *[a.copy() for a in axes], origin=box_centre + off)
        w = vox @ geo.image_to_world(sattr).T
        val = phantom.phantom_intensity(w[..., :3] - c, RADIUS)
        img = np.maximum(val * average / 0.55 + rng.normal(0.0, noise_sigma, val.shape), 0.0)
        out.append((img.astype(np.float32), sattr))
    return out, mask


def pvr_problem(name, workdir=None):
    """PVR4 / PVR8spx (module docstring): the stacks go through NIfTI files and the product's command line, which dumps what
    it would upload."""
    import subprocess
    import tempfile
    from . import build, nifti
    own = None
    if workdir is None:
        own = tempfile.TemporaryDirectory(prefix="svr_pvr_")
        workdir = own.name
    try:
        paths = []
        if name == "PVR4":
            stacks, mask = p4_stacks()
            for k, (d, a) in enumerate(stacks):
                nifti.write(os.path.join(workdir, f"s{k}.nii"), d, a)
                paths.append(os.path.join(workdir, f"s{k}.nii"))
            nifti.write(os.path.join(workdir, "mask.nii"), mask.data.astype(np.float32), mask.attr)
            opts = ["--thickness", *["2.5"] * 4, "--resolution", "1.0", "--patchSize", "32", "32", "--patchStride", "16", "16"]
            spx = False
        elif name == "PVR8spx":
            stacks, mask, mattr, rattr, rmask = phantom.make_stacks(8, (256, 256, 64), 1.0, 2.5, 2.5, 1.0, 100.0, seed=7, orientations=("ax", "cor", "sag"),
                                                                    stack_motion_mm=0.0, stack_motion_deg=0.0)
            for k, st in enumerate(stacks):
                nifti.write(os.path.join(workdir, f"s{k}.nii"), st.data, st.attr)
                paths.append(os.path.join(workdir, f"s{k}.nii"))
            nifti.write(os.path.join(workdir, "mask.nii"), rmask.astype(np.float32), rattr)
            opts = ["--thickness", *["2.5"] * 8, "--resolution", "0.5", "-s", "--spxSize", "32", "--spxExtend", "2"]
            spx = True
        else:
            raise ValueError(name)
        dump = os.path.join(workdir, "problem.bin")
        build.build()
        r = subprocess.run([build.PVR_CLI, "-o", os.path.join(workdir, "x.nii"), "-i", *paths, "-m", os.path.join(workdir, "mask.nii"), *opts,
                            "--no_registration", "--dumpProblem", dump, "--dryRun"], capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError("PVRreconstructionGPU --dryRun failed: " + r.stderr[-2000:])
        P = load_pvr_dump(dump, spx)
        P.name = name
        P.pvr = True
        return P
    finally:
        if own is not None:
            own.cleanup()


def p4_truth(prob, average=700.0, mask_path=MASK_FIXTURE):
    """the analytic phantom of problem_p4 sampled at the voxel centres of the problem's volume grid (what a perfect reconstruction of
    the noise-free, motion-corrected stacks would hold, up to the stacks' intensity scale): float32 [vz * vy * vx]"""
    _, c = load_bundled_mask(mask_path)
    vx, vy, vz = (int(v) for v in prob.vsize)
    kk, jj, ii = np.meshgrid(np.arange(vz), np.arange(vy), np.arange(vx), indexing="ij")
    i2w = np.asarray(prob.recon_i2w, np.float64).reshape(4, 4)
    w = np.stack([ii, jj, kk, np.ones_like(ii)], -1).astype(np.float64) @ i2w.T
    return (phantom.phantom_intensity(w[..., :3] - c, RADIUS) * average / 0.55).astype(np.float32).reshape(-1)


def get(name, **kw):
    """workload by the name bench.py / the tests use"""
    if name in ("PVR4", "PVR8spx"):
        return pvr_problem(name, **kw)
    if name == "P4":
        return problem_p4(**kw)
    if name == "P4s":
        return phantom.problem_p4(**kw)
    if name == "S8":
        return phantom.problem_s8(**kw)
    if name == "S8h":
        return phantom.make_problem(8, (256, 256, 64), 1.0, 2.5, 2.5, 0.5, 100.0, orientations=("ax", "cor", "sag"), name="S8h", **kw)
    raise ValueError(f"unknown workload {name}")
