"""Builds the HIP engine in-tree: fetalreconstruction_amd/lib/libsvr_hip.so (gfx950 only).

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "svr_hip.hip")
SRC_REG = os.path.join(HERE, "csrc", "svr_reg.inc")       # GPU registration, #included by svr_hip.hip
SRC_PYR = os.path.join(HERE, "csrc", "svr_pyr.inc")       # device pyramid of the IRTK registration, #included by svr_hip.hip
SRC_HOST = os.path.join(HERE, "csrc", "svr_host.cpp")     # plain host C++ (the irtkReconstruction mirror)
SRC_IO = os.path.join(HERE, "csrc", "svr_io.cpp")         # NIfTI-1 reader / writer (zlib)
SRC_PVR_HOST = os.path.join(HERE, "csrc", "pvr_host.cpp")  # the irtkPatchBasedReconstruction loop (host C++)
SRC_RCCL = os.path.join(HERE, "csrc", "svr_rccl.cpp")     # the collectives on RCCL (dlopen: no link-time dependency)
SRC_IRTK = os.path.join(HERE, "csrc", "irtk_reg.cpp")      # the IRTK registration schedule around the NCC cost (host C++)
SRC_PREP = os.path.join(HERE, "csrc", "svr_prep.h")       # pre-processing shared by the two command lines
SRC_SLIC = os.path.join(HERE, "csrc", "svr_slic.h")
SRC_CELL = os.path.join(HERE, "csrc", "svr_cell.inc")     # the scatter without atomics (cell-owned planes + combine), #included by svr_hip.hip
SRC_SORT = os.path.join(HERE, "csrc", "svr_sort.hip")     # radix sort / prefix sum from hipCUB for its work lists (own translation unit)
SRC_EM = os.path.join(HERE, "csrc", "svr_em.inc")         # the slice-level EM on the device, #included by svr_hip.hip
SRC_REGUL = os.path.join(HERE, "csrc", "svr_regul.inc")     # the fused volume update (Prep + edge-preserving regulariser), #included by svr_hip.hip
SRC_TILE = os.path.join(HERE, "csrc", "svr_tile.inc")     # the tile kernels of rounds 1-2 (fallbacks, the table's gather on coarse slices), #included by svr_hip.hip
SRC_SMALL = os.path.join(HERE, "csrc", "svr_small.inc")   # list compaction, reductions, EM / volume / bias kernels, the NCC cost, #included by svr_hip.hip
SRC_RCCL_ABI = os.path.join(HERE, "csrc", "svr_rccl_abi.h")   # the hand-written slice of rccl.h (checked by tests/rccl_abi_check.cpp)
SRC_SHARD = os.path.join(HERE, "csrc", "svr_shard.h")     # unit ranges + the one exchange per step, shared by the two host objects       # SLICO superpixel patches of the PVR command line
INC = os.path.join(os.path.dirname(HERE), "include", "svr_hip.h")
INC_HOST = os.path.join(os.path.dirname(HERE), "include", "svr_host.h")
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "libsvr_hip.so")
SRC_CLI = os.path.join(HERE, "csrc", "svr_cli.cpp")       # the SVRreconstructionGPU command line (host C++)
SRC_PVR_CLI = os.path.join(HERE, "csrc", "pvr_cli.cpp")   # the PVRreconstructionGPU command line (host C++)
BIN_DIR = os.path.join(HERE, "bin")
CLI = os.path.join(BIN_DIR, "SVRreconstructionGPU")
PVR_CLI = os.path.join(BIN_DIR, "PVRreconstructionGPU")

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # the canonical PSF sequence must be evaluated exactly as written (oracle parity)
    "-ffp-contract=off", "-fno-fast-math",
    # hardware global_atomic_add_f32 for the scatter (coarse-grained hipMalloc memory)
    "-munsafe-fp-atomics",
]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in (SRC, SRC_REG, SRC_PYR, SRC_HOST, SRC_IO, SRC_PVR_HOST, SRC_IRTK, SRC_RCCL, SRC_PREP, SRC_SLIC, SRC_SHARD, SRC_REGUL, SRC_EM, SRC_CELL, SRC_TILE, SRC_SMALL, SRC_RCCL_ABI, SRC_SORT, SRC_CLI, SRC_PVR_CLI, INC, INC_HOST,
                                               __file__)) or not (os.path.exists(CLI) and os.path.exists(PVR_CLI))


def build(force=False, verbose=False, extra=(), variant=None):
    """variant: tuning builds `lib/libsvr_hip_<variant>.so` with extra -D flags (tools/exp_*.py pick one
    through the SVR_HIP_LIB environment variable); the product is always the un-suffixed library."""
    out = OUT if not variant else os.path.join(OUT_DIR, f"libsvr_hip_{variant}.so")
    if not variant and not force and not needs_build():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [hipcc(), *FLAGS, *extra, "-o", out, SRC, SRC_SORT, SRC_HOST, SRC_PVR_HOST, SRC_IRTK, SRC_IO, SRC_RCCL, "-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if not variant:
        build_cli(verbose)
    return out


def build_cli(verbose=False):
    """SVRreconstructionGPU and PVRreconstructionGPU: plain host C++ linked against the engine library (found through its rpath)."""
    os.makedirs(BIN_DIR, exist_ok=True)
    cxx = shutil.which("g++") or hipcc()
    for exe, src in ((CLI, SRC_CLI), (PVR_CLI, SRC_PVR_CLI)):
        cmd = [cxx, "-O2", "-std=c++17", "-pthread", "-o", exe, src, "-L" + OUT_DIR, "-lsvr_hip", "-Wl,-rpath,$ORIGIN/../lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    variant = None
    extra = ["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else []
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        variant = sys.argv[i + 1]
        extra += [a for a in sys.argv[i + 2:] if a.startswith("-D")]
    print(build(force="--force" in sys.argv, verbose=True, extra=extra, variant=variant))
