"""Loader of `bin/PVRreconstructionGPU --dumpProblem <file> --dryRun` (csrc/pvr_cli.cpp): what the C++ command line is about to
hand to the engine, as a phantom.Problem the Python bindings can upload (engine.sync_gpu)."""
import numpy as np

from fetalreconstruction_amd import geometry as geo
from fetalreconstruction_amd.phantom import Problem


def load(path, superpixel):
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
    prob = Problem(vsize=(vx, vy, vz), vdim=tuple(float(v) for v in gdim), recon_i2w=gi2w.copy(), recon_w2i=gw2i.copy(), mask=mask.copy(),
                   slices=patches.copy(), slice_i2w=i2w.copy(), slice_w2i=w2i.copy(), slice_t=t.copy(), slice_tinv=ti.copy(),
                   slice_dim=dims.copy(), sizes_x=np.full(ns, px, np.int32), sizes_y=np.full(ns, py, np.int32),
                   stack_index=np.repeat(np.arange(nst, dtype=np.int32), counts), psf_c0=geo.psf_centre_offset(tuple(float(v) for v in gdim)),
                   min_intensity=float(vmin), max_intensity=float(vmax), name="pvr-dump")
    prob.patches_per_stack = [int(c) for c in counts]
    prob.spx_masks = spx
    return prob
