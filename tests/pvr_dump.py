"""Loader of `bin/PVRreconstructionGPU --dumpProblem <file> --dryRun` (csrc/pvr_cli.cpp); lives in the package now
(workloads.load_pvr_dump: bench.py's PVR workloads use it)."""
from fetalreconstruction_amd.workloads import load_pvr_dump


def load(path, superpixel):
    return load_pvr_dump(path, superpixel)
