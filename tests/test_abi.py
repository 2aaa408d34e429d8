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
