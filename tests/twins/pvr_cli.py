"""`PVRreconstructionGPU`-style command line over the MI355X engine (patch-to-volume reconstruction).
Mirrors main() of source/reconstructionGPU2/patchBasedReconMain.cpp ("pvrmain", options :108-131) and
irtkPatchBasedReconstruction<T>::run (irtkPatchBasedReconstruction.cpp = "PBR.cpp" :193-593): binarise the mask,
crop the stacks to it, resample the mask to the isotropic voxel size, match the stack intensities
(PBR.cpp:656-790: the target is the mean of all positive voxels), create the template (PBR.cpp:941-965: the
template stack's grid resampled, no extra slices), extract the patches, then `iterations + 1` passes of
Gaussian reconstruction -> robust statistics -> `sr_iterations` SR iterations.

    python -m tests.twins.pvr_cli -o recon.nii.gz -i s1.nii.gz s2.nii.gz -m mask.nii.gz \\
        [--patchSize 32 32] [--patchStride 16 16] [--resolution 0.75] [--iterations 7] [--sr_iterations 7]

The stack-to-stack registration (irtkStack3D3DRegistration, PBR.cpp:280-285) runs through csrc/irtk_reg.cpp with every
similarity on the GPU; between the outer passes every patch is registered to the volume with the same schedule
(patchBased2D3DRegistration<T>::runHybrid, what PBR.cpp:472-476 calls); --no_registration (not a reference option) skips both.  Not built: the CPU
path (--useCPU) and the evaluation options.  --resample resamples the cropped stacks to the output voxel size with IRTK's cubic
B-spline interpolator; --packages p_1 .. p_N splits every stack into its interleaved packages (PBR.cpp:134-146); --dilateMask n dilates the mask n times (26-connectivity); --existingReconTarget starts from a given volume and its grid, --hierarchical runs
iterations + 1 levels of shrinking patches (pvrmain:359-432).  --useFullSlices makes every slice one patch
(patchBasedObject.cuh:183-189).  -s/--superpixel cuts SLICO superpixel patches
(slic.py) instead of square ones; the patch-to-volume registration is skipped in that mode (undefined in the reference).
"""
from __future__ import annotations

import argparse
import copy
import sys

import numpy as np

from fetalreconstruction_amd import engine, nifti
from . import pvr
from fetalreconstruction_amd import geometry as geo
from fetalreconstruction_amd import preprocess as pp
from .cli import _load_transformation


def _parser():
    p = argparse.ArgumentParser(prog="PVRreconstructionGPU (MI355X)", description=__doc__.split("\n\n")[0])
    p.add_argument("-o", "--output", required=True)
    p.add_argument("-m", "--mask", required=True)
    p.add_argument("-i", "--input", nargs="+", required=True)
    p.add_argument("-t", "--transformation", nargs="+")
    p.add_argument("--patchSize", nargs=2, type=int, default=[32, 32])
    p.add_argument("--patchStride", nargs=2, type=int, default=[16, 16])
    p.add_argument("--resolution", type=float, default=0.75)
    p.add_argument("--noMatchIntensities", action="store_true")
    p.add_argument("--no_registration", action="store_true")
    p.add_argument("--iterations", type=int, default=7)
    p.add_argument("--sr_iterations", type=int, default=7)
    p.add_argument("--thickness", nargs="+", type=float)
    p.add_argument("-d", "--devices", nargs="+", type=int, default=[0])
    p.add_argument("-s", "--superpixel", action="store_true")
    p.add_argument("--spxSize", type=int, default=16)
    p.add_argument("--spxExtend", type=int, default=50)
    p.add_argument("--useFullSlices", action="store_true")
    p.add_argument("--hierarchical", action="store_true")
    p.add_argument("--existingReconTarget")
    p.add_argument("--dilateMask", type=int, default=0)
    p.add_argument("--packages", nargs="+", type=int)
    p.add_argument("--resample", action="store_true")
    p.add_argument("--useCPU", action="store_true", help=argparse.SUPPRESS)
    return p


def resample_attr(a, d):
    """irtkResampling::Initialize (irtkResampling.cc:74-130): int(n * d_old / d) voxels of size d, same origin."""
    r = copy.copy(a)
    n = [int(a.nx * a.dx / d), int(a.ny * a.dy / d), int(a.nz * a.dz / d)]
    size = [d, d, d]
    for k, old in enumerate((a.dx, a.dy, a.dz)):
        if n[k] < 1:
            n[k], size[k] = 1, old
    r.nx, r.ny, r.nz = n
    r.dx, r.dy, r.dz = size
    return r


def match_stack_intensities_pvr(stacks, transformations, mask, together=False):
    """PBR.cpp:656-790: like the SVR one, but the target average is the mean of all positive voxels of all stacks
    and a stack's own average counts only its positive voxels inside the mask."""
    total, count = np.float32(0), 0                      # m_average_value is a float accumulated voxel by voxel
    for s in stacks:
        pos = s.data[s.data > 0].astype(np.float32)
        if pos.size:
            total = np.cumsum(np.concatenate([[total], pos]), dtype=np.float32)[-1]
            count += pos.size
    average_value = float(np.float32(total / np.float32(count))) if count else 0.0
    m_w2i, ma = geo.world_to_image(mask.attr), mask.attr
    averages = []
    for st, t in zip(stacks, transformations):
        q = geo.apply_points(m_w2i, geo.apply_points(np.asarray(t, np.float64), geo.apply_points(geo.image_to_world(st.attr), pp._grid(st.attr))))
        idx = pp._round_half_away(q[..., :3]).astype(np.int64)
        ok = ((idx[..., 0] >= 0) & (idx[..., 0] < ma.nx) & (idx[..., 1] >= 0) & (idx[..., 1] < ma.ny) &
              (idx[..., 2] >= 0) & (idx[..., 2] < ma.nz))
        mv = mask.data[np.clip(idx[..., 2], 0, ma.nz - 1), np.clip(idx[..., 1], 0, ma.ny - 1),
                       np.clip(idx[..., 0], 0, ma.nx - 1)]
        sel = ok & (mv == 1) & (st.data > 0)
        if not sel.any():
            raise SystemExit("a stack has no overlap with the ROI")
        averages.append(float(st.data[sel].sum() / sel.sum()))
    glob = float(np.mean(averages))
    for st, av in zip(stacks, averages):
        f = average_value / (glob if together else av)
        st.data = np.where(st.data > 0, st.data * f, st.data)
    return average_value


def split_packages(stack, packages):
    """patchBasedPackageSplitter<T>::makePackageVolumes (patchBasedPackageSplitter.cpp:76-146): package l holds slices l, l + packages,
    ... of the stack at `packages` times the slice spacing, with its first voxel where slice l's first voxel was."""
    out = []
    a = stack.attr
    pkg_z = a.nz // packages
    i2w = geo.image_to_world(a)
    for l in range(packages):
        pa = copy.copy(a)
        pa.nz = pkg_z + 1 if pkg_z * packages + l < a.nz else pkg_z
        pa.dz = a.dz * packages
        pa.origin = np.asarray(a.origin, np.float64).copy()
        target = geo.apply_points(i2w, np.array([0.0, 0.0, float(l), 1.0]))
        first = geo.apply_points(geo.image_to_world(pa), np.array([0.0, 0.0, 0.0, 1.0]))
        pa.origin = pa.origin + (target - first)[:3]
        out.append(pp.Image(stack.data[l::packages][:pa.nz].copy(), pa))
    return out


def _bspline_coefficients(c):
    """ConvertToInterpolationCoefficients along the last axis, cubic spline, mirror boundaries
    (irtkBSplineInterpolateImageFunction.cc:68-144), in place."""
    import math
    n = c.shape[-1]
    if n == 1:
        return
    z = math.sqrt(3.0) - 2.0
    c *= (1.0 - z) * (1.0 - 1.0 / z)
    horizon = int(math.ceil(math.log(np.finfo(np.float64).eps) / math.log(abs(z))))
    zn = z
    if horizon < n:
        s = c[..., 0].copy()
        for k in range(1, horizon):
            s += zn * c[..., k]
            zn *= z
        c[..., 0] = s
    else:
        iz = 1.0 / z
        z2n = math.pow(z, float(n - 1))
        s = c[..., 0] + z2n * c[..., n - 1]
        z2n *= z2n * iz
        for k in range(1, n - 1):
            s += (zn + z2n) * c[..., k]
            zn *= z
            z2n *= iz
        c[..., 0] = s / (1.0 - zn * zn)
    for k in range(1, n):
        c[..., k] += z * c[..., k - 1]
    c[..., n - 1] = (z / (z * z - 1.0)) * (z * c[..., n - 2] + c[..., n - 1])
    for k in range(n - 2, -1, -1):
        c[..., k] = z * (c[..., k + 1] - c[..., k])


def resample_bspline(img, d):
    """irtkResampling<T> with irtkBSplineInterpolateImageFunction (cubic, clamped to the input's range), what --resample does to
    every cropped stack (PBR.cpp:225-246; irtkResampling.cc:74-175, irtkBSplineInterpolateImageFunction.cc:146-433).  The
    voxels are float, as in irtkGenericImage<float>."""
    a = img.attr
    coeff = np.array(img.data, np.float64)
    for axis in (2, 1, 0):                                                   # x, then y, then z
        v = np.moveaxis(coeff, axis, -1)
        _bspline_coefficients(v)
    out_attr = resample_attr(a, d)
    m = geo.mat_mul(geo.world_to_image(a), geo.image_to_world(out_attr))
    kk, jj, ii = np.meshgrid(np.arange(out_attr.nz), np.arange(out_attr.ny), np.arange(out_attr.nx), indexing="ij")
    p = [m[r, 0] * ii + m[r, 1] * jj + m[r, 2] * kk + m[r, 3] for r in range(3)]
    idx, wgt = [], []
    for x, n in zip(p, (a.nx, a.ny, a.nz)):
        i0 = np.floor(x).astype(np.int64) - 1
        w = x - (i0 + 1)
        w3 = (1.0 / 6.0) * w * w * w
        w0 = (1.0 / 6.0) + (1.0 / 2.0) * w * (w - 1.0) - w3
        w2 = w + w0 - 2.0 * w3
        w1 = 1.0 - w0 - w2 - w3
        wgt.append((w0, w1, w2, w3))
        half = 2 * n - 2
        ids = []
        for mm in range(4):
            q = i0 + mm
            if n == 1:
                q = np.zeros_like(q)
            else:
                q = np.where(q < 0, -q - half * ((-q) // half), q - half * (q // half))   # C division of non-negative operands
                q = np.where(q >= n, half - q, q)
            ids.append(q)
        idx.append(ids)
    value = np.zeros(p[0].shape, np.float64)
    for k in range(4):
        for j in range(4):
            for i in range(4):
                value += wgt[0][i] * wgt[1][j] * wgt[2][k] * coeff[idx[2][k], idx[1][j], idx[0][i]]
    value = np.clip(value, img.data.min(), img.data.max())
    return pp.Image(value.astype(np.float32).astype(np.float64), out_attr)


def dilate_mask(m, iterations):
    """irtkDilation<T> with CONNECTIVITY_26 (irtkDilation.cc:50-78), `iterations` runs: an interior voxel becomes the maximum of
    its 26 neighbours and itself, the voxels on the faces of the image keep their value."""
    m = np.asarray(m, np.float64)
    for _ in range(int(iterations)):
        out = m.copy()
        if min(m.shape) > 2:
            core = m[1:-1, 1:-1, 1:-1].copy()
            nz, ny, nx = m.shape
            for dz in (0, 1, 2):
                for dy in (0, 1, 2):
                    for dx in (0, 1, 2):
                        np.maximum(core, m[dz:nz - 2 + dz, dy:ny - 2 + dy, dx:nx - 2 + dx], out=core)
            out[1:-1, 1:-1, 1:-1] = core
        m = out
    return m


def prepare(stacks, transformations, mask, resolution, template, no_match, register=None, dilate=0, resample=False):
    """PBR.cpp:197-310.  `register(stacks, transformations, iso_mask) -> transformations` is the stack-to-stack registration
    (irtkStack3D3DRegistration, :280-285) or None.  Returns (stacks, transformations, iso mask, template attributes, recon mask)."""
    mask = pp.Image((np.trunc(mask.data) != 0).astype(np.float64), mask.attr)            # :201-209, (unsigned int) cast
    if dilate:
        mask = pp.Image(dilate_mask(mask.data, dilate), mask.attr)                        # :212-223
    for k in range(len(stacks)):                                                          # :229-236
        m = pp.TransformMask(stacks[k].attr, mask, transformations[k])
        stacks[k] = pp.CropImage(stacks[k], m)
        if resample:
            stacks[k] = resample_bspline(stacks[k], resolution)                           # :237-246
    iso_mask = pp.transform_nn(mask, resample_attr(mask.attr, resolution))               # :258-266
    if register is not None and len(stacks) > 1:
        transformations = register(stacks, transformations, iso_mask)                    # :280-285
    if not no_match:
        match_stack_intensities_pvr(stacks, transformations, iso_mask)                   # :291
    tattr = resample_attr(stacks[template].attr, resolution)                              # CreateTemplate :941-965
    recon_mask = pp.TransformMask(tattr, iso_mask, transformations[template])            # :303-304
    return stacks, transformations, iso_mask, tattr, recon_mask


def _hip_engine(prob, device):
    rec = engine.Reconstruction(device)                  # raises when the HIP library is missing: no CPU path
    rec.set_option("pvr", 1)
    engine.sync_gpu(rec, prob, quality_factor=1.0)       # m_quality_factor = 1 (PBR.cpp:415)
    if getattr(prob, "spx_masks", None) is not None:
        rec.set_spx_masks(prob.spx_masks)                # ImagePatch2D::spxMask of every patch
    return rec


def main(argv=None, _engine_factory=_hip_engine, _ncc_backend=None):
    """`_engine_factory` / `_ncc_backend` exist for the CPU tests, which drive the same pipeline over the test oracle."""
    a = _parser().parse_args(argv)
    if a.useCPU:
        raise SystemExit("--useCPU is not supported by this build: there is no CPU reconstruction path (see fetalreconstruction_amd/pvr_cli.py)")
    n = len(a.input)
    stacks = []
    for path in a.input:
        d, at = nifti.read(path)
        stacks.append(pp.Image(d.astype(np.float64), at))
    specs = a.transformation or ["id"] * n
    ts = [_load_transformation(s) for s in specs]
    thickness = a.thickness or [2.0 * s.attr.dz for s in stacks]                          # pvrmain: twice the z spacing
    template = next((k for k, s in enumerate(specs) if s == "id"), 0)
    if a.packages and len(a.packages) == n:                                               # setImageStacks, PBR.cpp:134-146: every package
        if min(a.packages) < 1:                                                           # becomes a stack of its own; m_template_num
            raise SystemExit("--packages takes positive integers")                        # keeps indexing the new list
        split = [(pk, t, th) for s, t, th, k in zip(stacks, ts, thickness, a.packages) for pk in split_packages(s, k)]
        stacks, ts, thickness = [list(v) for v in zip(*split)]
        n = len(stacks)
    md, mat = nifti.read(a.mask)
    def register(st, tr, iso):                                                            # irtkStack3D3DRegistration<T>::run
        from fetalreconstruction_amd import host
        rec = engine.Reconstruction(a.devices[0]) if _engine_factory is _hip_engine else None
        out, evals = host.StackRegistrations(rec, [s.data for s in st], [s.attr for s in st], tr, template, mask=iso.data,
                                             mask_attr=iso.attr, keep_origin=True, backend=_ncc_backend)
        print(f"stack-to-stack registration: {evals} similarity evaluations", file=sys.stderr)
        return list(out)

    stacks, ts, iso_mask, tattr, recon_mask = prepare(stacks, ts, pp.Image(md.astype(np.float64), mat), a.resolution, template,
                                                      a.noMatchIntensities, None if a.no_registration else register, a.dilateMask, a.resample)
    pstacks = [pvr.Stack(s.data.astype(np.float32), s.attr, t, th / 2.0) for s, t, th in zip(stacks, ts, thickness)]
    if a.superpixel and a.useFullSlices:
        raise SystemExit("--superpixel with --useFullSlices is not supported by this build")
    existing = None
    if a.existingReconTarget:                                                             # setExistingReconstructionTarget :185-191
        ed, tattr = nifti.read(a.existingReconTarget)                                     # the volume and its grid (no CreateTemplate)
        existing = ed.astype(np.float32)
        recon_mask = pp.TransformMask(tattr, iso_mask, ts[template])
    pos = np.concatenate([s.data[s.data > 0].astype(np.float32) for s in stacks])          # computeMinMaxIntensities :792-814:
    vmin, vmax = float(pos.min()), float(pos.max())                                       # over the whole (cropped) stacks

    def run_level(psize, pstride, iterations, existing):
        """irtkPatchBasedReconstruction<T>::run from the patch extraction on (the set-up above gives the same result every time)."""
        prob = pvr.make_pvr_problem(pstacks, iso_mask.data, iso_mask.attr, tattr, recon_mask.data, psize, pstride,
                                    superpixel=a.superpixel, full_slices=a.useFullSlices)
        print(f"{n} stacks, {prob.ns} patches of {prob.slices.shape[2]}x{prob.slices.shape[1]} {prob.patches_per_stack}, volume {prob.vsize} at "
              f"{a.resolution} mm", file=sys.stderr)
        rec = _engine_factory(prob, a.devices[0])
        drv = pvr.irtkPatchBasedReconstruction(rec, prob.patches_per_stack, vmin, vmax)
        T = np.stack([np.asarray(ts[int(k)], np.float64) for k in prob.stack_index])      # the registrators' m_transformations
        for it in range(iterations + 1):                                                  # PBR.cpp:445
            have_volume = it > 0 or existing is not None                                  # :456
            if have_volume and not a.no_registration and a.superpixel:
                # runHybrid registers the square CPU patches of generatePatchesCPU (patchBased2D3DRegistration.cpp:227-375) and
                # updateTransformationMatrices then reads one transformation per GPU patch from that shorter list: undefined in
                # the reference for superpixel patches, not done here
                print("superpixel mode: the patch-to-volume registration is skipped", file=sys.stderr)
            elif have_volume and not a.no_registration:                                   # PBR.cpp:452-489 (runHybrid)
                T, evals = drv.PatchToVolumeRegistration(prob, T, tattr, backend=_ncc_backend, volume=existing if it == 0 else None)
                print(f"patch-to-volume registration: {evals} similarity evaluations", file=sys.stderr)
            drv.reconstruct_iteration(a.sr_iterations)
            print(f"iteration {it}: sigma {float(drv.m_sigma_gpu):.4g} mix {float(drv.m_mix_gpu):.3f}", file=sys.stderr)
        return np.asarray(rec.syncCPU(), np.float32).reshape(tattr.nz, tattr.ny, tattr.nx).copy()

    psize, pstride = list(a.patchSize), list(a.patchStride)
    if a.superpixel:                                                                      # pvrmain:291-296
        psize, pstride = [a.spxSize, a.spxSize], [a.spxExtend, a.spxExtend]
    if a.hierarchical and a.useFullSlices:                                                # pvrmain:282-285 "SVR ON"
        a.hierarchical = False
    if not a.hierarchical:
        out = run_level(psize, pstride, a.iterations, existing)
    else:
        # pvrmain:359-432: iterations + 1 levels of one registration-reconstruction iteration each; a level starts from the volume
        # of the level before (its "reconimage1_<size>_<stride>.nii.gz") and cuts patches 4 pixels smaller (stride 2 smaller, not
        # for superpixels)
        for level in range(a.iterations + 1):
            if min(psize + pstride) < 1:
                raise SystemExit(f"hierarchical mode: the patch size reached zero at level {level} (--patchSize - 4 * --iterations must stay positive)")
            print(f"hierarchical level {level}: patch size {psize[0]} stride {pstride[0]}", file=sys.stderr)
            existing = out = run_level(psize, pstride, 1, existing)
            psize = [psize[0] - 4, psize[1] - 4]
            if not a.superpixel:
                pstride = [pstride[0] - 2, pstride[1] - 2]
    nifti.write(a.output, out, tattr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
