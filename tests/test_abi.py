"""The C-ABI library loads and exports every symbol include/svr_hip.h declares (no compute)."""
import ctypes
import os
import re

from fetalreconstruction_amd import build as svr_build
from fetalreconstruction_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="svr_hip.h", prefix="svr_"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_all_declared_symbols():
    path = svr_build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_lists_every_declared_symbol():
    assert sorted(engine.EXPORTS) == _declared()


def test_host_object_exports_all_declared_symbols():
    from fetalreconstruction_amd import host
    lib = ctypes.CDLL(svr_build.build())
    names = _declared("svr_host.h", "svrh_")
    assert len(names) >= 15 and sorted(host.HOST_EXPORTS + host.IRTK_EXPORTS) == names
    assert not [n for n in names if not hasattr(lib, n)]
    pv = _declared("svr_host.h", "pvrh_")
    assert sorted(host.PVR_HOST_EXPORTS) == pv and not [n for n in pv if not hasattr(lib, n)]
    lib.pvrh_create.restype = ctypes.c_void_p
    assert lib.pvrh_create(None, None, 0, ctypes.c_float(0), ctypes.c_float(1)) is None
    assert not [n for n in host.IRTK_EXPORTS if not hasattr(lib, n)]
    io = [n for n in _declared("svr_host.h", "svr_") if not n.startswith("svrh_") and n != "svr_collectives"]
    assert sorted(host.IO_EXPORTS + host.COMM_EXPORTS) == sorted(n for n in io if hasattr(lib, n)) == sorted(io)
    lib.svr_comm_create.restype = ctypes.c_void_p
    assert lib.svr_comm_create(0, 2, None, None) is None       # no id / engine -> refused, librccl is not even opened
    lib.svrh_create.restype = ctypes.c_void_p
    assert lib.svrh_create(None, 4, 0, 4, None) is None        # no engine -> refused


def test_create_rejects_bad_arguments_without_gpu():
    lib = engine.load_library()
    assert lib.svr_create(0, None) == 10001          # SVR_E_ARG
    assert lib.svr_last_error(None) == b"null context"
    assert lib.svr_volume_voxels(None) == 0
    assert lib.svr_device_ptr(None, 0) is None


def test_host_threads_follow_the_override_and_the_cpu_quota():
    """svr_host_threads(): the affinity mask cut to the cgroup CPU quota; SVR_HOST_THREADS overrides (read once per process)."""
    import subprocess, sys, os
    code = ("from fetalreconstruction_amd import engine; print(engine.load_library().svr_host_threads())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    plain = int(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, check=True).stdout)
    assert 1 <= plain <= len(os.sched_getaffinity(0))
    forced = int(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, check=True,
                                env={**os.environ, "SVR_HOST_THREADS": "3"}).stdout)
    assert forced == 3


def test_the_hand_written_slice_of_rccl_h_matches_the_image_s_header(tmp_path):
    """csrc/svr_rccl.cpp binds librccl by dlsym against prototypes and enum values written out by hand (csrc/svr_rccl_abi.h: the library must
    build and load without RCCL).  tests/rccl_abi_check.cpp puts them next to /opt/rocm/include/rccl/rccl.h and static_asserts that they
    agree -- the first real N > 1 run should not be the first time anyone finds out.  With a control: the same check must FAIL to compile
    once a value of the hand-written header is changed."""
    import shutil
    import subprocess
    import pytest
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr) or not shutil.which("g++"):
        pytest.skip("no RCCL header / compiler in this image")
    src = os.path.join(ROOT, "tests", "rccl_abi_check.cpp")
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
    ok = subprocess.run(cmd + [src], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr[-3000:]
    # the control: a changed enum value, and a changed prototype, each break the compilation
    abi = open(os.path.join(ROOT, "fetalreconstruction_amd", "csrc", "svr_rccl_abi.h")).read()
    chk = open(src).read().replace('#include "../fetalreconstruction_amd/csrc/svr_rccl_abi.h"', '#include "svr_rccl_abi.h"')
    for old, new, what in (("ncclFloat32 = 7", "ncclFloat32 = 6", "ncclDataType_t values"),
                           ("typedef int (*AllGather_fn)(const void *, void *, size_t, int, ncclComm_t, hipStream_t);",
                            "typedef int (*AllGather_fn)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);", "ncclAllGather")):
        assert old in abi
        d = tmp_path / what.replace(" ", "_")
        d.mkdir()
        (d / "svr_rccl_abi.h").write_text(abi.replace(old, new))
        (d / "check.cpp").write_text(chk)
        bad = subprocess.run(cmd + ["-I" + str(d), str(d / "check.cpp")], capture_output=True, text=True)
        assert bad.returncode != 0 and what in bad.stderr, (what, bad.stderr[-1500:])
