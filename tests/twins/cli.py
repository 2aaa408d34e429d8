"""`SVRreconstructionGPU`-style command line over the MI355X engine: NIfTI stacks + mask in,
reconstructed volume out.  Mirrors main() of source/reconstructionGPU2/reconstruction.cc ("main.cc"):
option names and defaults of main.cc:164-211, the set-up of main.cc:386-815 (template, mask, cropping,
intensity matching, slices, masking) and the registration-reconstruction loop of main.cc:816-1237 with its
smoothing schedule.

    python -m tests.twins.cli -o recon.nii.gz -i s1.nii.gz s2.nii.gz s3.nii.gz -m mask.nii.gz \\
        [--thickness 2.5 2.5 2.5] [--resolution 0.75] [--iterations 4] [--useGPUReg]

Stack transformations (`id`, IRTK rigid `dof` files or 4x4 text matrices) start the stack-to-stack registration
(StackRegistrations, before and after the other stacks are cropped, main.cc:661,711); slice-to-volume registration is
the reference's default IRTK schedule with every similarity evaluated on the GPU (csrc/irtk_reg.cpp) or, with
--useGPUReg, the reference's GPU registration.  --no_registration (not a reference option) skips both.  --packages runs PackageToVolume
with the schedule of main.cc:832-864; --tfolder reads `transformation<i>.dof` per slice (--debug writes them next to the output).  Superpixels and the CPU reconstruction path are refused, loudly.
"""
from __future__ import annotations

import argparse
import sys

import numpy as np

from fetalreconstruction_amd import engine, host, nifti
from fetalreconstruction_amd import preprocess as pp
from . import registration as reg
from .reconstruction import irtkReconstruction


def _bool(v):
    v = str(v).lower()
    if v in ("1", "true", "yes", "on"):
        return True
    if v in ("0", "false", "no", "off"):
        return False
    raise argparse.ArgumentTypeError(f"boolean expected, got {v}")


def _parser():
    p = argparse.ArgumentParser(prog="SVRreconstructionGPU (MI355X)", description=__doc__.split("\n\n")[0])
    p.add_argument("-o", "--output", required=True)
    p.add_argument("-m", "--mask")
    p.add_argument("-i", "--input", nargs="+", required=True)
    p.add_argument("-t", "--transformation", nargs="+")
    p.add_argument("--thickness", nargs="+", type=float)
    p.add_argument("--iterations", type=int, default=4)
    p.add_argument("--sigma", type=float, default=12.0)
    p.add_argument("--resolution", type=float, default=0.75)
    p.add_argument("--multires", type=int, default=3)
    p.add_argument("--average", type=float, default=700.0)
    p.add_argument("--delta", type=float, default=150.0)
    p.add_argument("--lambda", dest="lam", type=float, default=0.02)
    p.add_argument("--lastIterLambda", type=float, default=0.01)
    p.add_argument("--smooth_mask", type=float, default=4.0)
    # po::value<bool> options of the reference take a value (--debug 1); a bare flag is accepted too.  The value of
    # --no_intensity_matching lands in `intensity_matching` (main.cc:186): 0 switches the matching off, a bare flag does as well
    p.add_argument("--no_intensity_matching", nargs="?", const="0", default=None, type=_bool)
    p.add_argument("--num_stacks_tuner", type=int, default=0)
    for ignored in ("--log_prefix", "--low_intensity_cutoff", "--patchSize", "--patchStride"):     # no log files; bias / patch modes are off
        p.add_argument(ignored, help=argparse.SUPPRESS)
    for ignored in ("--no_log", "--global_bias_correction"):
        p.add_argument(ignored, nargs="?", const="1", type=_bool, help=argparse.SUPPRESS)
    p.add_argument("--useCPUReg", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--debug_gpu", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--force_exclude", nargs="+", type=int, default=[])
    p.add_argument("--rec_iterations_first", type=int, default=4)
    p.add_argument("--rec_iterations_last", type=int, default=13)
    p.add_argument("--useGPUReg", action="store_true")
    p.add_argument("--no_registration", action="store_true")
    p.add_argument("--disableBiasCorrection", action="store_true", default=True)
    p.add_argument("-d", "--devices", nargs="+", type=int, default=[0])
    p.add_argument("--debug", nargs="?", const="1", default=False, type=_bool)
    p.add_argument("--packages", nargs="+", type=int)
    p.add_argument("--tfolder")
    for refused in ("--useCPU", "--patchBased", "--superpixelBased", "--sfolder"):
        p.add_argument(refused, nargs="*", help=argparse.SUPPRESS)
    return p


def _load_transformation(spec):
    if spec == "id":
        return np.eye(4)
    try:
        return nifti.read_dof(spec)[1]                  # IRTK rigid dof file
    except engine.SvrError:
        pass
    m = np.loadtxt(spec, dtype=np.float64)
    if m.shape != (4, 4):
        raise SystemExit(f"transformation {spec}: expected 'id', an IRTK rigid dof file or a 4x4 text matrix")
    return m


def main(argv=None):
    a = _parser().parse_args(argv)
    for refused in ("useCPU", "patchBased", "superpixelBased", "sfolder"):
        if getattr(a, refused) is not None:
            raise SystemExit(f"--{refused} is not supported by this build (see fetalreconstruction_amd/cli.py)")
    if a.num_stacks_tuner > 0:                                              # main.cc:406-419: only the first stacks are used
        k = a.num_stacks_tuner
        a.input = a.input[:k]
        for name in ("transformation", "thickness", "packages"):
            if getattr(a, name):
                setattr(a, name, getattr(a, name)[:k])
    n = len(a.input)
    stacks = []
    for path in a.input:                                                    # main.cc:386-430
        d, at = nifti.read(path)
        if d.ndim != 3:
            raise SystemExit(f"{path}: 3-D stacks expected")
        stacks.append(pp.Image(d.astype(np.float64), at))
    transformations = [_load_transformation(s) for s in (a.transformation or ["id"] * n)]
    if len(transformations) != n:
        raise SystemExit("one transformation per stack expected")
    thickness = a.thickness or [2.0 * s.attr.dz for s in stacks]           # main.cc:422-431: twice the z spacing
    if len(thickness) != n:
        raise SystemExit("one thickness per stack expected")
    if a.packages and len(a.packages) != n:
        raise SystemExit("one package count per stack expected")
    template = next((k for k, s in enumerate(a.transformation or ["id"] * n) if s == "id"), None)   # first 'id' stack
    if template is None:
        raise SystemExit("Please identify the template by assigning id transformation.")           # main.cc:452-457
    if a.mask:
        md, mat = nifti.read(a.mask)
        mask = pp.Image(md.astype(np.float64), mat)
    else:
        # no mask given: CreateMask(stacks[templateNumber]) binarises the template stack (> 0), in case it was padded; the
        # normal mask path follows (main.cc:458-480, RG.cc:736-748)
        mask = pp.Image((stacks[template].data > 0).astype(np.float64), stacks[template].attr)

    # template stack: mask onto its grid, crop (main.cc:583-584)
    if mask is not None:
        m = pp.TransformMask(stacks[template].attr, mask, transformations[template])
        stacks[template] = pp.CropImage(stacks[template], m)
    tattr, resolution = pp.CreateTemplate(stacks[template].attr, a.resolution)                   # main.cc:607
    vol_mask = pp.SetMask(tattr, mask, a.smooth_mask)                                            # main.cc:610
    rec = engine.Reconstruction(a.devices[0])

    def stack_registrations(ts):                                                                 # StackRegistrations, RG.cc:849-1001
        if a.no_registration or n < 2:
            return ts
        out, evals = host.StackRegistrations(rec, [s.data for s in stacks], [s.attr for s in stacks], ts, template,
                                             mask=vol_mask.data if mask is not None else None, mask_attr=vol_mask.attr)
        print(f"stack-to-stack registration: {evals} similarity evaluations", file=sys.stderr)
        return list(out)

    transformations = stack_registrations(transformations)                                       # main.cc:657-662
    for k in range(n):                                                                           # main.cc:676-700
        if k == template:
            continue
        m = pp.TransformMask(stacks[k].attr, vol_mask, transformations[k])
        stacks[k] = pp.CropImage(stacks[k], m)
    transformations = stack_registrations(transformations)                                       # main.cc:707-713
    factors = pp.MatchStackIntensitiesWithMasking(stacks, transformations, vol_mask, a.average,
                                                  together=a.no_intensity_matching is not None and not a.no_intensity_matching)              # main.cc:676-679
    slices, attrs, slice_t, stack_index = pp.CreateSlicesAndTransformations(stacks, transformations, thickness)
    slices = pp.MaskSlices(slices, attrs, slice_t, vol_mask)                                     # main.cc:700
    prob = pp.build_problem(tattr, vol_mask, slices, attrs, slice_t, stack_index)
    print(f"{n} stacks, {prob.ns} slices of up to {prob.slices.shape[2]}x{prob.slices.shape[1]}, volume {prob.vsize} "
          f"at {resolution} mm, stack factors {np.round(factors, 3)}", file=sys.stderr)

    engine.sync_gpu(rec, prob)                                                                   # SyncGPU, main.cc:722
    drv = irtkReconstruction(rec, prob.ns, max_intensity=prob.max_intensity, min_intensity=prob.min_intensity)
    # `--no_intensity_matching v` stores v in intensity_matching (main.cc:183): 0 (and the bare flag here) switches Bias / Scale off
    drv._intensity_matching = a.no_intensity_matching is None or bool(a.no_intensity_matching)     # main.cc:1018, 1062
    drv.SetForceExcludedSlices(a.force_exclude)
    rs = reg.PrepareRegistrationSlices(rec, prob.slices, prob.slice_attr, resolution) if a.useGPUReg else None
    T = np.stack(slice_t)
    if a.tfolder:                                                                                # ReadTransformation, RG.cc:4733-4765
        import os
        T = np.stack([nifti.read_dof(os.path.join(a.tfolder, f"transformation{i}.dof"))[1] for i in range(prob.ns)])
        ti = np.stack([np.linalg.inv(t) for t in T])
        rec.SetSliceMatrices(np.stack([t.astype(np.float32).reshape(16) for t in T]), np.stack([t.astype(np.float32).reshape(16) for t in ti]),
                             prob.slice_i2w, prob.slice_w2i, prob.slice_i2w, prob.slice_w2i, prob.recon_i2w, prob.recon_w2i)
    for it in range(a.iterations):                                                               # main.cc:816-1237
        slice_reg = it > 0 and not a.no_registration
        if slice_reg and a.packages and it <= a.iterations * (a.multires - 1) // a.multires and it < a.iterations - 1:
            # packages first (main.cc:832-864): plain, even/odd, even/odd halves; from iteration 4 on also the slices
            vol = rec.syncCPU().reshape(tattr.nz, tattr.ny, tattr.nx)
            T, evals = host.PackageToVolume(rec, [s.data for s in stacks], [s.attr for s in stacks], a.packages, T, tattr, vol,
                                            evenodd=it >= 2, half=it >= 3, half_iter=max(1, it - 2) if it >= 4 else 1)
            print(f"package-to-volume registration: {evals} similarity evaluations", file=sys.stderr)
            slice_reg = it >= 4
            if not slice_reg:
                ti = np.stack([np.linalg.inv(t) for t in T])
                rec.SetSliceMatrices(np.stack([t.astype(np.float32).reshape(16) for t in T]),
                                     np.stack([t.astype(np.float32).reshape(16) for t in ti]), prob.slice_i2w, prob.slice_w2i,
                                     prob.slice_i2w, prob.slice_w2i, prob.recon_i2w, prob.recon_w2i)
        if slice_reg:                                                                            # main.cc:829-880
            if a.useGPUReg:
                T = reg.SliceToVolumeRegistrationGPU(rec, rs, T)
            else:                                                                                # SliceToVolumeRegistration, RG.cc:2291-2303
                vol = rec.syncCPU().reshape(tattr.nz, tattr.ny, tattr.nx)                        # _reconstructed after SyncCPU, main.cc:1189
                T, evals = host.SliceToVolumeRegistration(rec, prob.slices, prob.slice_attr, T, tattr, vol)
                print(f"slice-to-volume registration: {evals} similarity evaluations", file=sys.stderr)
            ti = np.stack([np.linalg.inv(t) for t in T])
            rec.SetSliceMatrices(np.stack([t.astype(np.float32).reshape(16) for t in T]),
                                 np.stack([t.astype(np.float32).reshape(16) for t in ti]), prob.slice_i2w, prob.slice_w2i,
                                 prob.slice_i2w, prob.slice_w2i, prob.recon_i2w, prob.recon_w2i)  # UpdateGPUTranformationMatrices
        if it == a.iterations - 1:                                                               # main.cc:884-896
            drv.SetSmoothingParameters(a.delta, a.lastIterLambda)
        else:
            lam = a.lam
            for i in range(a.multires):
                if it == a.iterations * (a.multires - i - 1) // a.multires:
                    drv.SetSmoothingParameters(a.delta, lam)
                lam *= 2
        (drv.SpeedupOn if it < a.iterations - 1 else drv.SpeedupOff)()
        rec_it = a.rec_iterations_last if it == a.iterations - 1 else a.rec_iterations_first    # main.cc:1001-1012
        drv.reconstruct_iteration(rec_it)
        print(f"iteration {it}: sigma {drv._sigma_gpu:.4g} mix {drv._mix_gpu:.3f} "
              f"excluded slices {int((drv._slice_weight_gpu < 0.5).sum())}", file=sys.stderr)
    rec.RestoreSliceIntensities(factors, prob.stack_index)                                       # main.cc:1189-1193
    drv.ScaleVolumeGPU()
    out = rec.syncCPU().reshape(tattr.nz, tattr.ny, tattr.nx)
    nifti.write(a.output, out, tattr)
    if a.debug:                                                                                  # SaveTransformations, RG.cc:4903-4915
        import os
        folder = os.path.dirname(os.path.abspath(a.output))
        for i, t in enumerate(T):
            nifti.write_dof(os.path.join(folder, f"transformation{i}.dof"), host.irtk_rigid_parameters(t)[0])
    return 0


if __name__ == "__main__":
    sys.exit(main())
