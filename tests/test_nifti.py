"""NIfTI-1 I/O of the engine library (csrc/svr_io.cpp) with the reference's image conventions
(irtkFileNIFTIToImage.cc:168-345, irtkImageToFileNIFTI.cc:65-145).  No GPU needed."""
import gzip
import struct

import numpy as np
import pytest

from fetalreconstruction_amd import geometry as geo
from fetalreconstruction_amd import nifti
from fetalreconstruction_amd.engine import SvrError


def _attr():
    r = geo.rigid_matrix(rx=20, ry=-35, rz=50)[:3, :3]
    return geo.ImageAttributes(7, 5, 3, 1.2, 0.9, 2.5, r[:, 0], r[:, 1], r[:, 2], origin=np.array([3.0, -4.5, 10.25]))


def _header(dim, pixdim, datatype, bitpix, endian="<", qform=None, sform=None, slope=0.0, inter=0.0, vox_offset=352.0):
    """A raw 348-byte NIfTI-1 header built field by field from the standard's layout."""
    h = bytearray(348)
    struct.pack_into(endian + "i", h, 0, 348)
    struct.pack_into(endian + "8h", h, 40, *dim)
    struct.pack_into(endian + "hh", h, 70, datatype, bitpix)
    struct.pack_into(endian + "8f", h, 76, *pixdim)
    struct.pack_into(endian + "fff", h, 108, vox_offset, slope, inter)
    if qform is not None:
        struct.pack_into(endian + "h", h, 252, 1)
        struct.pack_into(endian + "6f", h, 256, *qform)
    if sform is not None:
        struct.pack_into(endian + "h", h, 254, 1)
        struct.pack_into(endian + "12f", h, 280, *np.asarray(sform, np.float32).reshape(-1)[:12])
    h[344:348] = b"n+1\0"
    return bytes(h) + b"\0\0\0\0"


@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
def test_write_read_round_trip(tmp_path, ext):
    a = _attr()
    d = np.random.default_rng(0).normal(size=(3, 5, 7)).astype(np.float32)
    p = tmp_path / ("img" + ext)
    nifti.write(p, d, a)
    r, ra = nifti.read(p)
    assert np.array_equal(r, d)
    assert (ra.nx, ra.ny, ra.nz) == (7, 5, 3)
    assert np.allclose(geo.image_to_world(ra), geo.image_to_world(a), atol=2e-6)       # float32 header fields
    assert np.allclose(ra.origin, a.origin, atol=2e-6)
    raw = (gzip.open(p) if ext.endswith("gz") else open(p, "rb")).read()
    assert len(raw) == 352 + d.nbytes and raw[344:347] == b"n+1"
    assert struct.unpack_from("<i", raw, 0)[0] == 348 and struct.unpack_from("<f", raw, 108)[0] == 352.0
    assert struct.unpack_from("<hh", raw, 252) == (1, 0)                                 # qform only (irtkNIFTI.h:135-145)
    assert struct.unpack_from("<h", raw, 70)[0] == 16                                    # float32


def test_large_volumes_are_deflated_in_parallel_as_gzip_members(tmp_path):
    """A volume of 16 MiB or more is written as a sequence of gzip members (csrc/svr_io.cpp): zlib's gzread (the C++ reader)
    and Python's gzip module read it as one stream, voxels and header unchanged."""
    a = geo.ImageAttributes(260, 200, 90, 0.5, 0.5, 0.5, origin=np.array([3.0, -2.0, 7.0]))
    rng = np.random.default_rng(5)
    v = (rng.random((90, 200, 260)) * 100).astype(np.float32)
    v[:30] = 0
    path = tmp_path / "big.nii.gz"
    nifti.write(path, v, a)
    w, b = nifti.read(path)
    assert np.array_equal(w, v)
    assert (b.nx, b.ny, b.nz) == (a.nx, a.ny, a.nz) and np.allclose(b.origin, a.origin, atol=1e-5)
    raw = gzip.open(path).read()
    assert len(raw) == 352 + v.size * 4
    assert np.array_equal(np.frombuffer(raw[352:], np.float32).reshape(v.shape), v)
    assert open(path, "rb").read().count(b"\x1f\x8b\x08\x00") >= 4              # 18.7 MB of payload in 4 MiB members


def test_left_handed_axes_use_qfac(tmp_path):
    a = _attr()
    a.zaxis = -a.zaxis                                                                   # det < 0
    d = np.arange(105, dtype=np.float32).reshape(3, 5, 7)
    nifti.write(tmp_path / "l.nii", d, a)
    raw = open(tmp_path / "l.nii", "rb").read()
    assert struct.unpack_from("<f", raw, 76)[0] == -1.0                                  # pixdim[0] = qfac
    r, ra = nifti.read(tmp_path / "l.nii")
    assert np.allclose(ra.zaxis, a.zaxis, atol=1e-6) and np.allclose(ra.origin, a.origin, atol=2e-6)


def test_sform_only_big_endian_int16_with_scaling(tmp_path):
    a = _attr()
    m = geo.image_to_world(a)
    vals = np.arange(105, dtype=np.int16).reshape(3, 5, 7) - 20
    raw = _header((3, 7, 5, 3, 1, 1, 1, 1), (1, 1.2, 0.9, 2.5, 1, 1, 1, 1), 4, 16, endian=">", sform=m[:3], slope=0.5,
                  inter=10.0) + vals.astype(">i2").tobytes()
    p = tmp_path / "s.nii"
    p.write_bytes(raw)
    d, ra = nifti.read(p)
    assert np.array_equal(d, vals.astype(np.float32) * 0.5 + 10.0)                       # scl_slope / scl_inter
    assert np.allclose(geo.image_to_world(ra), m, atol=2e-6)
    # origin = world position of the centre voxel (irtkFileNIFTIToImage.cc:309-325)
    assert np.allclose(ra.origin, (m @ np.array([3.0, 2.0, 1.0, 1.0]))[:3], atol=2e-6)


def test_no_transform_gives_the_default_radiological_frame(tmp_path):
    raw = _header((3, 4, 3, 2, 1, 1, 1, 1), (1, 2.0, 3.0, 4.0, 1, 1, 1, 1), 2, 8) + bytes(range(24))
    p = tmp_path / "d.nii"
    p.write_bytes(raw)
    d, ra = nifti.read(p)
    assert d.shape == (2, 3, 4) and d[1, 2, 3] == 23
    assert np.allclose(ra.xaxis, [-1, 0, 0]) and np.allclose(ra.yaxis, [0, 1, 0]) and np.allclose(ra.zaxis, [0, 0, 1])
    assert np.allclose(ra.origin, 0)                                                     # :262-289: centred on (0,0,0)
    assert (ra.dx, ra.dy, ra.dz) == (2.0, 3.0, 4.0)


def test_quaternion_of_a_half_turn(tmp_path):
    a = geo.ImageAttributes(4, 4, 2, 1, 1, 1, np.array([-1.0, 0, 0]), np.array([0, -1.0, 0]), np.array([0, 0, 1.0]),
                            origin=np.array([1.0, 2.0, 3.0]))                           # rotation by pi about z: a = 0
    d = np.ones((2, 4, 4), np.float32)
    nifti.write(tmp_path / "h.nii", d, a)
    _, ra = nifti.read(tmp_path / "h.nii")
    assert np.allclose(geo.image_to_world(ra), geo.image_to_world(a), atol=2e-6)


def test_errors_are_loud(tmp_path):
    with pytest.raises(SvrError):
        nifti.read(tmp_path / "missing.nii")
    (tmp_path / "junk.nii").write_bytes(b"\0" * 400)
    with pytest.raises(SvrError):
        nifti.read(tmp_path / "junk.nii")
    raw = _header((3, 4, 3, 2, 1, 1, 1, 1), (1, 1, 1, 1, 1, 1, 1, 1), 16, 32) + b"\0" * 10     # truncated data
    (tmp_path / "short.nii").write_bytes(raw)
    with pytest.raises(SvrError):
        nifti.read(tmp_path / "short.nii")
    with pytest.raises(SvrError):
        nifti.write(tmp_path / "x.nii", np.zeros((2, 2, 2), np.float32), _attr())               # shape mismatch


def test_irtk_dof_files(tmp_path):
    """Rigid dof: big-endian {815007, 2, 6} + six doubles (irtkRigidTransformation.cc:392-451)."""
    p = [1.5, -2.25, 3.0, 10.0, -20.0, 30.0]
    nifti.write_dof(tmp_path / "t.dof", p)
    raw = (tmp_path / "t.dof").read_bytes()
    assert len(raw) == 60 and struct.unpack(">III", raw[:12]) == (815007, 2, 6)
    assert struct.unpack(">6d", raw[12:]) == tuple(p)
    q, m = nifti.read_dof(tmp_path / "t.dof")
    assert np.array_equal(q, p) and np.allclose(m, geo.rigid_matrix(*p), atol=1e-15)
    (tmp_path / "t.dof.gz").write_bytes(gzip.compress(raw))                    # irtkCifstream reads through zlib
    assert np.array_equal(nifti.read_dof(tmp_path / "t.dof.gz")[0], p)
    (tmp_path / "bad.dof").write_bytes(struct.pack(">III", 815007, 3, 12) + b"\\0" * 48)
    with pytest.raises(SvrError):
        nifti.read_dof(tmp_path / "bad.dof")


# ---- pinned against the reference's own NIfTI-1 library (oracle/_ref/libnifti_ref.so) -------------------
# oracle/Makefile compiles source/IRTKSimple2/nifti/{niftilib/nifti1_io.c, znzlib/znzlib.c} where they lie
# (plain C + system zlib) with the flat-argument harness oracle/nifti_ref_driver.c.
def _ref():
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libnifti_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libnifti_ref.so not built (needs /root/reference; `make -C oracle`)")
    return C.CDLL(path), C


def _ref_read(path, with_data=True, dtype=np.float32):
    lib, C = _ref()
    dims, pix, codes = (C.c_int * 5)(), (C.c_float * 4)(), (C.c_int * 4)()
    qto, sto, scl = (C.c_float * 16)(), (C.c_float * 16)(), (C.c_float * 2)()
    assert lib.ref_nifti_read(str(path).encode(), dims, pix, codes, qto, sto, scl, None, 0) == 0
    n = dims[1] * dims[2] * dims[3] * max(dims[4], 1)
    buf = np.zeros(n * codes[3], np.uint8)
    if with_data:
        assert lib.ref_nifti_read(str(path).encode(), dims, pix, codes, qto, sto, scl, buf.ctypes.data_as(C.c_void_p),
                                  C.c_long(buf.nbytes)) == 0
    return dict(dims=list(dims), pix=list(pix), qform=codes[0], sform=codes[1], datatype=codes[2],
                qto=np.array(qto[:]).reshape(4, 4), sto=np.array(sto[:]).reshape(4, 4), scl=list(scl),
                data=buf.view(dtype) if with_data else None)


def _ref_write(path, data, pix, datatype, qto=None, sto=None, slope=1.0, inter=0.0, nt=1):
    lib, C = _ref()
    d = np.ascontiguousarray(data)
    nz, ny, nx = d.shape[-3:]
    f16 = lambda m: None if m is None else (C.c_float * 16)(*np.asarray(m, np.float32).reshape(-1))   # noqa: E731
    rc = lib.ref_nifti_write(str(path).encode(), (C.c_int * 3)(nx, ny, nz), nt, (C.c_float * 3)(*pix), datatype, f16(qto), f16(sto),
                             C.c_float(slope), C.c_float(inter), d.ctypes.data_as(C.c_void_p))
    assert rc == 0


@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
@pytest.mark.parametrize("left_handed", [False, True])
def test_written_files_parse_in_the_reference_library(tmp_path, ext, left_handed):
    """svr_nifti_write -> nifti_image_read of the reference: the header irtkNIFTIHeader::Initialize would produce
    (irtkNIFTI.h:84-160: float32, qform_code 1 = the image-to-world matrix, sform_code 0)."""
    a = _attr()
    if left_handed:
        a.yaxis = -a.yaxis
    d = np.random.default_rng(1).normal(size=(3, 5, 7)).astype(np.float32)
    p = tmp_path / ("w" + ext)
    nifti.write(p, d, a)
    r = _ref_read(p)
    assert r["dims"] == [3, 7, 5, 3, 1] and r["datatype"] == 16 and (r["qform"], r["sform"]) == (1, 0)
    assert np.allclose(r["pix"][:3], [1.2, 0.9, 2.5], atol=1e-6) and r["pix"][3] == (-1.0 if left_handed else 1.0)
    assert np.allclose(r["qto"], geo.image_to_world(a), atol=2e-5)              # through the float32 quaternion
    assert np.array_equal(r["data"].reshape(3, 5, 7), d)
    assert r["scl"][0] in (0.0, 1.0) and r["scl"][1] == 0.0


def test_quaternion_matches_the_reference_library(tmp_path):
    lib, C = _ref()
    for seed, flip in ((0, False), (1, True), (2, False)):
        rng = np.random.default_rng(seed)
        rot = geo.rigid_matrix(rx=rng.uniform(-170, 170), ry=rng.uniform(-80, 80), rz=rng.uniform(-170, 170))[:3, :3]
        a = geo.ImageAttributes(6, 4, 5, 0.8, 1.3, 2.0, rot[:, 0], rot[:, 1], -rot[:, 2] if flip else rot[:, 2],
                                origin=rng.uniform(-50, 50, 3))
        nifti.write(tmp_path / "q.nii", np.zeros((5, 4, 6), np.float32), a)
        raw = open(tmp_path / "q.nii", "rb").read()
        mine = struct.unpack_from("<6f", raw, 256)                               # quatern b c d, qoffset x y z
        qfac = struct.unpack_from("<f", raw, 76)[0]
        q, out = (C.c_float * 10)(), (C.c_float * 16)()
        lib.ref_quatern_round_trip((C.c_float * 16)(*geo.image_to_world(a).astype(np.float32).reshape(-1)), out, q)
        assert np.allclose(mine, q[:6], atol=2e-6) and qfac == q[9]
        assert np.allclose(np.array(out[:]).reshape(4, 4), geo.image_to_world(a), atol=2e-5)


@pytest.mark.parametrize("case", ["int16_scaled_gz", "uint8", "float64", "sform_only", "both_forms", "float32_4d"])
def test_reference_written_files_read_back(tmp_path, case):
    """nifti_image_write of the reference -> svr_nifti_read: voxels (with scl_slope / scl_inter) and the geometry rule of
    irtkFileNIFTIToImage::ReadHeader (qform first, then sform; axes = matrix columns / voxel size, origin = centre)."""
    a = _attr()
    m = geo.image_to_world(a)
    rng = np.random.default_rng(3)
    other = geo.image_to_world(geo.ImageAttributes(7, 5, 3, 1.2, 0.9, 2.5, origin=np.array([1.0, 2.0, 3.0])))
    pix = (1.2, 0.9, 2.5)
    kw, nt, expect_m = dict(qto=m), 1, m
    if case == "int16_scaled_gz":
        raw, dt, path = rng.integers(-3000, 3000, (3, 5, 7)).astype(np.int16), 4, tmp_path / "r.nii.gz"
        kw.update(slope=0.5, inter=-7.0)
        expect = raw.astype(np.float32) * np.float32(0.5) + np.float32(-7.0)
    elif case == "uint8":
        raw, dt, path = rng.integers(0, 255, (3, 5, 7)).astype(np.uint8), 2, tmp_path / "r.nii"
        expect = raw.astype(np.float32)
    elif case == "float64":
        raw, dt, path = rng.normal(size=(3, 5, 7)), 64, tmp_path / "r.nii"
        expect = raw.astype(np.float32)
    elif case == "sform_only":
        raw, dt, path = rng.normal(size=(3, 5, 7)).astype(np.float32), 16, tmp_path / "r.nii"
        kw, expect = dict(sto=m), raw
    elif case == "both_forms":
        raw, dt, path = rng.normal(size=(3, 5, 7)).astype(np.float32), 16, tmp_path / "r.nii.gz"
        kw, expect = dict(qto=m, sto=other), raw                                  # the qform wins
    else:
        raw, dt, path, nt = rng.normal(size=(2, 3, 5, 7)).astype(np.float32), 16, tmp_path / "r.nii", 2
        expect = raw
    _ref_write(path, raw, pix, dt, nt=nt, **kw)
    d, ra = nifti.read(path)
    assert d.shape == expect.shape and np.allclose(d, expect, rtol=1e-6, atol=1e-6)
    assert (ra.nx, ra.ny, ra.nz) == (7, 5, 3) and np.allclose([ra.dx, ra.dy, ra.dz], pix, atol=1e-6)
    assert np.allclose(geo.image_to_world(ra), expect_m, atol=3e-5)
