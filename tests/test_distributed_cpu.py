"""N>1 path on CPU: world-size-2 gloo run of the sharded driver (slice sharding, volume-pair
all-reduce, M-step scalar all-reduce, per-slice all-gathers) with the oracle as each rank's engine,
against the single-process run of the same problem."""
import os
import socket
import tempfile

import numpy as np
import pytest

from fetalreconstruction_amd import phantom
from fetalreconstruction_amd.sharding import shard_slices
from tests.twins.reconstruction import irtkReconstruction


def _problem():
    return phantom.make_problem(2, (18, 18, 3), 1.2, 2.4, None, 1.0, 11.5, seed=4, orientations=("ax", "cor"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir, slabs=False, iters=1):
    import torch.distributed as dist
    from fetalreconstruction_amd.sharding import TorchComm
    from oracle import pyoracle as po
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        P = _problem()
        act = (P.slices != -1).reshape(P.ns, -1).sum(1)
        lo, hi = shard_slices(act, world)[rank]
        eng = po.OracleReconstruction(phantom.sub_problem(P, lo, hi), po.CANON)
        drv = irtkReconstruction(eng, P.ns, (lo, hi), TorchComm(slabs=slabs), P.max_intensity, P.min_intensity)
        drv.SetSmoothingParameters(150, 0.02)
        drv.reconstruct_iteration(iters)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), recon=eng.recon, scale=drv._scale_gpu,
                 sw=drv._slice_weight_gpu, em=np.array([drv._sigma_gpu, drv._mix_gpu, drv._m_gpu, drv._mix_s_gpu]),
                 lohi=np.array([lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world_size_2_matches_single_process(oracle_mod):
    import torch.multiprocessing as mp
    P = _problem()
    eng = oracle_mod.OracleReconstruction(P, oracle_mod.CANON)
    ref = irtkReconstruction(eng, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
    ref.SetSmoothingParameters(150, 0.02)
    ref.reconstruct_iteration(1)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0, r1 = np.load(os.path.join(d, "rank0.npz")), np.load(os.path.join(d, "rank1.npz"))
    assert r0["lohi"][0] == 0 and r0["lohi"][1] == r1["lohi"][0] and r1["lohi"][1] == P.ns   # every slice kept
    # both ranks end with the same volume and the same host state ...
    assert np.array_equal(r0["recon"], r1["recon"])
    assert np.array_equal(r0["scale"], r1["scale"]) and np.array_equal(r0["sw"], r1["sw"])
    # ... which is the single-process result up to the float32 rounding of the per-rank partial sums
    scale = np.abs(eng.recon).max()
    assert np.max(np.abs(r0["recon"] - eng.recon)) < 2e-5 * scale
    assert np.allclose(r0["scale"], ref._scale_gpu, rtol=1e-5)
    assert np.allclose(r0["sw"], ref._slice_weight_gpu, atol=1e-4)
    assert np.allclose(r0["em"], [ref._sigma_gpu, ref._mix_gpu, ref._m_gpu, ref._mix_s_gpu], rtol=1e-5)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_slab_update_is_the_replicated_update_bit_for_bit(world):
    """The volume update of a sharded run by z-slabs (csrc/svr_slab.inc; restated in numpy for the oracle engines:
    reconstruction.slab_plan_numpy / slab_update_numpy): reduce-scatter of addon | cmap at the mask's voxels of every rank's slab
    and its two halo planes, the rank's planes of Prep + regulariser, all-gather of the new volume at the dilated mask's voxels.
    Two SR iterations at world 2 and 3 (a middle slab has two halos): every rank ends with the volume of the run that
    all-reduces the pair and updates the whole volume on every rank -- bit for bit at world 2, where a + b = b + a; at world 3
    gloo's ring adds a voxel's three values in an order that depends on where the voxel sits in the message, and the two runs
    lay their messages out differently: float round-off of that one sum (the in-process group of the command lines adds in rank
    order in both forms: tests/test_preprocess.py::test_slab_update_gives_the_replicated_updates_bits, three ranks, bit for bit)."""
    import torch.multiprocessing as mp
    vols = {}
    for slabs in (False, True):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_worker, args=(world, _free_port(), d, slabs, 2), nprocs=world, join=True)
            rs = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]
            for r in rs[1:]:
                assert np.array_equal(rs[0]["recon"], r["recon"])
            vols[slabs] = (rs[0]["recon"].copy(), rs[0]["scale"].copy(), rs[0]["em"].copy())
    assert np.abs(vols[True][0]).max() > 0
    if world == 2:
        assert np.array_equal(vols[True][0], vols[False][0])
        assert np.array_equal(vols[True][1], vols[False][1]) and np.array_equal(vols[True][2], vols[False][2])
    else:
        assert np.array_equal(vols[True][0] == -1, vols[False][0] == -1) and np.array_equal(vols[True][0] != 0, vols[False][0] != 0)
        assert np.abs(vols[True][0] - vols[False][0]).max() <= 1e-6 * np.abs(vols[False][0]).max()
        assert np.allclose(vols[True][1], vols[False][1], rtol=1e-6) and np.allclose(vols[True][2], vols[False][2], rtol=1e-6)


def test_slab_plan_covers_every_voxel_once():
    """slab boundaries by mask-voxel count; reduce-scatter ranges = slab + one halo plane either side; all-gather ranges partition
    the dilated mask"""
    from tests.twins.reconstruction import slab_plan_numpy
    rng = np.random.default_rng(3)
    z, y, x = np.mgrid[:23, :17, :19]
    mask = (((z - 11) / 9.0) ** 2 + ((y - 8) / 6.5) ** 2 + ((x - 9) / 7.0) ** 2 < 1).astype(np.float32)
    mask[rng.integers(0, 23, 40), rng.integers(0, 17, 40), rng.integers(0, 19, 40)] = 0
    for world in (1, 2, 3, 5, 8, 23, 40):
        p = slab_plan_numpy(mask, world)
        zb = p["zb"]
        assert zb[0] == 0 and zb[-1] == 23 and all(a <= b for a, b in zip(zb, zb[1:]))
        cover = np.zeros(len(p["didx"]), int)
        for st, cnt in p["ag"]:
            cover[st:st + cnt] += 1
        assert (cover == 1).all()
        planes = p["midx"] // (17 * 19)
        for r, (st, cnt) in enumerate(p["rs"]):
            if zb[r] < zb[r + 1]:
                pl = planes[st:st + cnt]
                want = (planes >= zb[r] - 1) & (planes <= zb[r + 1])
                assert cnt == want.sum() and (len(pl) == 0 or (pl.min() >= zb[r] - 1 and pl.max() <= zb[r + 1]))
            else:
                assert cnt == 0
        counts = [((planes >= zb[r]) & (planes < zb[r + 1])).sum() for r in range(world)]
        if world <= 8:
            assert max(counts) - min(counts) <= 2 * (mask != 0).reshape(23, -1).sum(1).max()      # equal shares up to whole planes


# ---- the patch-to-volume loop sharded by patches ("PVR: identical with patches as the unit") ----------------
def _pvr_problem():
    from tests.twins import pvr
    stacks, mask, mattr, rattr, rmask = phantom.make_stacks(3, (24, 24, 4), 1.1, 2.2, None, 1.0, 11.0, seed=4, orientations=("ax", "sag", "cor"))
    return pvr.make_pvr_problem(stacks, mask, mattr, rattr, rmask, (16, 16), (8, 8))


def shard_patches(P, world):
    """contiguous patch ranges balanced by the pixels that carry data (what csrc/pvr_cli.cpp does)"""
    return shard_slices((P.slices > 0).reshape(P.ns, -1).sum(1), world)


def _pvr_worker(rank, world, port, outdir):
    import torch.distributed as dist
    from tests.twins import pvr
    from fetalreconstruction_amd.sharding import TorchComm
    from oracle import pyoracle as po
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        P = _pvr_problem()
        lo, hi = shard_patches(P, world)[rank]
        eng = po.OracleReconstruction(phantom.sub_problem(P, lo, hi), po.CANON, pvr=True)
        drv = pvr.irtkPatchBasedReconstruction(eng, P.patches_per_stack, P.min_intensity, P.max_intensity, patch_range=(lo, hi), comm=TorchComm())
        drv.reconstruct_iteration(2)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), recon=eng.recon, scale=drv.scale, pw=drv.patch_weight, pot=drv.patch_potential,
                 em=np.array([drv.m_sigma_gpu, drv.m_mix_gpu, drv.m_m_gpu, drv.m_mix_s_gpu]), lohi=np.array([lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_pvr_world_size_2_matches_single_process(oracle_mod):
    """Patches [lo, hi) per rank, balanced by data-carrying pixels, a stack boundary inside one rank's range: one all-reduce of
    recon|volw, one of addon|cmap per SR iteration, the robust-statistics / M-step scalars and the patch potentials in one
    exchange each -- and the reference's within-stack indexing of the potentials (patchBasedRobustStatistics_gpu.cu:256-276)
    applied to the GLOBAL patch numbering on every rank."""
    import torch.multiprocessing as mp
    from tests.twins import pvr
    P = _pvr_problem()
    eng = oracle_mod.OracleReconstruction(P, oracle_mod.CANON, pvr=True)
    ref = pvr.irtkPatchBasedReconstruction(eng, P.patches_per_stack, P.min_intensity, P.max_intensity)
    ref.reconstruct_iteration(2)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_pvr_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0, r1 = np.load(os.path.join(d, "rank0.npz")), np.load(os.path.join(d, "rank1.npz"))
    assert r0["lohi"][0] == 0 and r0["lohi"][1] == r1["lohi"][0] and r1["lohi"][1] == P.ns
    assert r0["lohi"][1] not in np.cumsum(P.patches_per_stack)         # the cut is not a stack boundary
    for k in ("recon", "scale", "pw", "pot", "em"):
        assert np.array_equal(r0[k], r1[k]), k                         # both ranks: the same volume and host state
    assert (r0["pot"][max(P.patches_per_stack):] == 0).all()           # the quirk, on global indices
    scale = np.abs(eng.recon).max()
    assert np.max(np.abs(r0["recon"] - eng.recon)) < 2e-5 * scale
    assert np.allclose(r0["scale"], ref.scale, rtol=1e-5) and np.allclose(r0["pw"], ref.patch_weight, atol=1e-4)
    assert np.allclose(r0["pot"], ref.patch_potential, rtol=1e-5, atol=1e-9)
    assert np.allclose(r0["em"], [ref.m_sigma_gpu, ref.m_mix_gpu, ref.m_m_gpu, ref.m_mix_s_gpu], rtol=1e-5)
