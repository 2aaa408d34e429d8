"""The C++ sharded host (csrc/svr_host.cpp + csrc/svr_shard.h + csrc/svr_slab.inc) where a collective is not a no-op: TWO PROCESSES on
the one GPU of the box, each with its own engine context and its spatially compact share of the slices (the r-th half of every stack,
sharding.shard_units), exchanging through gloo with host-staged, rank-ordered device collectives (host.py: the torch callbacks with
TorchComm(slabs=True); RCCL refuses two ranks on one device).  The closest thing to RCCL at world 2 this box can run: the scatter of a
rank's own slices, reduce-scatter -> the rank's slab of the volume update -> all-gather, the M-step's sums meeting on the device, one host
exchange per SR iteration -- against the one-rank run, and against the replicated form of the update bit for bit."""
import os
import socket
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu if __name__ != "__main__" else None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# the default suite: two processes on the fixed problem, three on a random one; a longer hunt: SVR_TWO_RANK_WORLD=8 SVR_TWO_RANK_SEED=5
# (recorded in round 5: worlds 2, 3, 4, 5 x seeds 1-6 and world 8 x seeds 1-3, all three forms of the update: profiles/r05_multi_process_hunt.txt)
CASES = [(int(os.environ["SVR_TWO_RANK_WORLD"]), os.environ.get("SVR_TWO_RANK_SEED"))] if "SVR_TWO_RANK_WORLD" in os.environ else [(2, None), (3, "1")]


def _problem(seed=None):
    from fetalreconstruction_amd import phantom
    if seed == "many":                                    # 1120 slices of 12 x 12 pixels (tests/test_round6_gpu.py: the arena of the small results)
        return phantom.make_problem(4, (12, 12, 280), 1.2, 0.25, 2.5, 1.0, 9.0, seed=5, orientations=("ax", "cor", "sag", "ax"), name="many-slices")
    if seed is None:
        return phantom.make_problem(3, (40, 36, 10), 1.1, 2.2, None, 1.0, 15.0, seed=11, orientations=("ax", "cor", "sag"), name="two-rank")
    rng = np.random.default_rng(int(seed))
    n = int(rng.integers(2, 5))
    return phantom.make_problem(n, (int(rng.integers(28, 44)), int(rng.integers(28, 44)), int(rng.integers(6, 12))), float(rng.uniform(0.9, 1.3)),
                                float(rng.uniform(1.8, 2.6)), None, float(rng.uniform(0.8, 1.2)), float(rng.uniform(12.0, 16.0)), seed=int(seed),
                                orientations=tuple(rng.choice(["ax", "cor", "sag"], n, replace=True)), name="two-rank-fuzz")


def _worker(rank, world, port, outdir, slabs, slab_update=True, seed=None):
    import torch                                       # before the engine's library: torch carries its own copy of the HIP runtime
    import torch.distributed as dist
    from fetalreconstruction_amd import engine as E, host, phantom
    from fetalreconstruction_amd.sharding import TorchComm, shard_units, slice_cost_weights
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        P = _problem(seed)
        act = (P.slices != -1).reshape(P.ns, -1).sum(1)
        work = slice_cost_weights(act, P.slice_i2w, P.slice_t, P.recon_w2i, P.slice_dim, P.vdim[0])
        order, ranges = shard_units(work, P.stack_index, world, "spatial")
        lo, hi = ranges[rank]
        rec = E.Reconstruction(0)
        E.sync_gpu(rec, phantom.sub_problem(P, 0, 0, select=order[lo:hi]))
        d = host.irtkReconstruction(rec, P.ns, (lo, hi), TorchComm(device=None, slabs=slabs), P.max_intensity, P.min_intensity)
        d.set_unit_order(order)
        if not slab_update:                               # the replicated form of the update with the same launcher (device collectives and all)
            d.set_slab_update(False)
        d.SetSmoothingParameters(150, 0.02)
        rec.timer_enable(True)
        d.reconstruct_iteration(3)
        st = d.state()
        tm = rec.timers()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), recon=rec.syncCPU(), scale=st["scale"], sw=st["slice_weight"], pot=st["slice_potential"],
                 em=np.array([st[k] for k in ("sigma", "mix", "m", "mean_s", "mean_s2", "sigma_s", "sigma_s2", "mix_s")]), order=order,
                 lohi=np.array([lo, hi]), counts=np.array([tm["reduce_scatter"][1], tm["allgather"][1], tm["allreduce"][1], tm["exchange_host"][1]]))
        rec.close()
    finally:
        dist.destroy_process_group()



def _spawn(world, slabs, slab_update, outdir, seed=None, kind="svr"):
    """two worker PROCESSES of this file (not torch.multiprocessing: importing torch into the pytest process next to the engine's library
    puts two HIP runtimes into one process, which corrupts the heap at exit)"""
    import subprocess
    import sys
    port = _free_port()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(world), str(port), outdir, str(int(slabs)), str(int(slab_update)), str(seed), kind],
                              cwd=root, env=dict(os.environ, PYTHONPATH=root), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("WORLD,seed", CASES)
def test_two_processes_on_one_gpu_through_the_cpp_sharded_host(WORLD, seed):
    from fetalreconstruction_amd import engine as E, host
    P = _problem(seed)
    rec = E.Reconstruction(0)
    E.sync_gpu(rec, P)
    ref = host.irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity)
    ref.SetSmoothingParameters(150, 0.02)
    ref.reconstruct_iteration(3)
    v_ref, s_ref = rec.syncCPU().copy(), ref.state()
    rec.close()
    runs = {}
    for key, slabs, slab_update in (("slab", True, True), ("replicated", True, False), ("no device collectives", False, True)):
        with tempfile.TemporaryDirectory() as d:
            _spawn(WORLD, slabs, slab_update, d, seed)
            runs[key] = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(WORLD)]
    for key, rr in runs.items():
        r0 = rr[0]
        slabs = key != "no device collectives"
        # all ranks end with the same volume and the same host state
        for r1 in rr[1:]:
            for k in ("recon", "scale", "sw", "pot", "em"):
                assert np.array_equal(r0[k], r1[k], equal_nan=True), (slabs, k)
        assert r0["lohi"][0] == 0 and all(rr[i]["lohi"][1] == rr[i + 1]["lohi"][0] for i in range(WORLD - 1)) and rr[-1]["lohi"][1] == P.ns
        # Gaussian pass: one all-reduce; per SR iteration: reduce-scatter + all-gather (slab) or one all-reduce (replicated), ONE host exchange
        rs, ag, ar, ex = (int(v) for v in r0["counts"])
        assert (rs, ag, ar) == ((3, 3, 1) if key == "slab" else (0, 0, 4)), (key, rs, ag, ar)
        # with the launcher's device collectives: ONE exchange in all (the robust statistics' two sums) -- the M-step's sums and the slice-level
        # EM's potentials meet on the device (csrc/svr_em.inc); without them: robust statistics + the first E-step + two per SR iteration
        assert ex == (1 if slabs else 2 + 2 * 3), (slabs, ex)
        # ... which is the one-rank result up to the float rounding of the per-rank partial sums; per-slice vectors in the sharded numbering
        order = r0["order"]
        assert np.abs(r0["recon"] - v_ref).max() <= 2e-5 * np.abs(v_ref).max()
        assert np.array_equal(r0["recon"] == -1, v_ref == -1)
        assert np.allclose(r0["scale"], s_ref["scale"][order], rtol=1e-5) and np.allclose(r0["sw"], s_ref["slice_weight"][order], atol=1e-4)
        assert np.allclose(r0["em"], [s_ref[k] for k in ("sigma", "mix", "m", "mean_s", "mean_s2", "sigma_s", "sigma_s2", "mix_s")], rtol=1e-4)
    # the slab update against the replicated one: at world 2 a sum of two terms has one order, so the two forms give the same bits whatever the
    # collectives do inside; beyond that the replicated form's all-reduce (gloo's here, RCCL's in the product) adds in an order of its own and the
    # two agree to the float-sum tolerance (the rank-ordered in-process group of `-d 0 0 0` and the numpy collectives of the CPU tests keep the
    # bits at world 3: tests/test_distributed_cpu.py)
    a, b = runs["slab"][0], runs["replicated"][0]
    if WORLD == 2:
        assert np.array_equal(a["recon"], b["recon"]) and np.array_equal(a["scale"], b["scale"]) and np.array_equal(a["sw"], b["sw"]) and np.array_equal(a["em"], b["em"])
    else:
        assert np.abs(a["recon"] - b["recon"]).max() <= 2e-5 * np.abs(v_ref).max() and np.allclose(a["sw"], b["sw"], atol=1e-4)


def _pvr_problem(seed):
    from fetalreconstruction_amd import phantom
    from tests.twins import pvr
    rng = np.random.default_rng(500 + int(seed))
    n = int(rng.integers(2, 4))
    stacks, mask, mattr, rattr, rmask = phantom.make_stacks(n, (int(rng.integers(24, 34)), int(rng.integers(24, 34)), int(rng.integers(4, 7))),
                                                            float(rng.uniform(0.9, 1.3)), float(rng.uniform(1.8, 2.6)), None, 1.0, float(rng.uniform(10.0, 13.0)),
                                                            seed=int(seed), orientations=tuple(rng.choice(["ax", "cor", "sag"], n, replace=True)))
    return pvr.make_pvr_problem(stacks, mask, mattr, rattr, rmask, (16, 16), (8, 8))


def _pvr_worker(rank, world, port, outdir, slabs, slab_update, seed):
    import torch
    import torch.distributed as dist
    from fetalreconstruction_amd import engine as E, host, phantom
    from fetalreconstruction_amd.sharding import TorchComm, patch_cost_weights, shard_units
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        P = _pvr_problem(seed)
        work = patch_cost_weights((P.slices > 0).reshape(P.ns, -1).sum(1), P.slice_i2w, P.slice_t, P.recon_w2i)
        order, ranges = shard_units(work, P.stack_index, world, "spatial")
        lo, hi = ranges[rank]
        rec = E.Reconstruction(0)
        rec.set_option("pvr", 1)
        E.sync_gpu(rec, phantom.sub_problem(P, 0, 0, select=order[lo:hi]), quality_factor=1.0)
        d = host.irtkPatchBasedReconstruction(rec, P.patches_per_stack, P.min_intensity, P.max_intensity, (lo, hi), TorchComm(device=None, slabs=slabs))
        d.set_unit_order(order)
        if not slab_update:
            d.set_slab_update(False)
        rec.timer_enable(True)
        d.reconstruct_iteration(3)
        st = d.state()
        tm = rec.timers()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), recon=rec.syncCPU(), scale=st["scale"], sw=st["patch_weight"], pot=st["patch_potential"],
                 em=np.array([st[k] for k in ("m_sigma_gpu", "m_mix_gpu", "m_m_gpu", "m_mean_s_gpu", "m_mean_s2_gpu", "m_sigma_s_gpu", "m_sigma_s2_gpu", "m_mix_s_gpu")]),
                 order=order, lohi=np.array([lo, hi]),
                 counts=np.array([tm["reduce_scatter"][1], tm["allgather"][1], tm["allreduce"][1], tm["exchange_host"][1]]))
        rec.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("WORLD,seed", [(int(os.environ["SVR_TWO_RANK_WORLD"]), os.environ.get("SVR_TWO_RANK_SEED", "1"))] if "SVR_TWO_RANK_WORLD" in os.environ
                         else [(2, "1"), (3, "2")])
def test_processes_on_one_gpu_through_the_cpp_sharded_patch_based_host(WORLD, seed):
    """The same for csrc/pvr_host.cpp: patches dealt spatially, the unit-order layer around the reference's within-stack indexing of the
    potentials, the slab update and the patch-level EM on the device (csrc/svr_em.inc, patch form), between PROCESSES; against the one-rank
    object: the volume to 1e-4 (float sums regrouped by rank and through the patch-level EM's weights), the same excluded patches, per-patch
    vectors that are the one-rank run's in the sharded numbering, one host exchange in the whole outer iteration."""
    from fetalreconstruction_amd import engine as E, host
    P = _pvr_problem(seed)
    rec = E.Reconstruction(0)
    rec.set_option("pvr", 1)
    E.sync_gpu(rec, P, quality_factor=1.0)
    ref = host.irtkPatchBasedReconstruction(rec, P.patches_per_stack, P.min_intensity, P.max_intensity)
    ref.reconstruct_iteration(3)
    v_ref, s_ref = rec.syncCPU().copy(), ref.state()
    rec.close()
    for key, slabs, slab_update in (("slab", True, True), ("replicated", True, False), ("no device collectives", False, True)):
        with tempfile.TemporaryDirectory() as d:
            _spawn(WORLD, slabs, slab_update, d, seed, kind="pvr")
            rr = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(WORLD)]
        r0 = rr[0]
        for r1 in rr[1:]:
            for k in ("recon", "scale", "sw", "pot", "em"):
                assert np.array_equal(r0[k], r1[k], equal_nan=True), (key, k)
        assert r0["lohi"][0] == 0 and all(rr[i]["lohi"][1] == rr[i + 1]["lohi"][0] for i in range(WORLD - 1)) and rr[-1]["lohi"][1] == P.ns
        rs, ag, ar, ex = (int(v) for v in r0["counts"])
        assert (rs, ag, ar) == ((3, 3, 1) if key == "slab" else (0, 0, 4)), (key, rs, ag, ar)
        assert ex == (1 if slabs else 2 + 2 * 3), (key, ex)            # with device collectives: the robust statistics' sums only
        order = r0["order"]
        assert np.abs(r0["recon"] - v_ref).max() <= 1e-4 * np.abs(v_ref).max() and np.array_equal(r0["recon"] > 0, v_ref > 0)
        assert np.array_equal(r0["pot"] == -1, s_ref["patch_potential"][order] == -1)
        assert np.allclose(r0["scale"], s_ref["scale"][order], rtol=2e-5) and np.allclose(r0["sw"], s_ref["patch_weight"][order], atol=2e-4)
        assert np.allclose(r0["em"], [s_ref[k] for k in ("m_sigma_gpu", "m_mix_gpu", "m_m_gpu", "m_mean_s_gpu", "m_mean_s2_gpu", "m_sigma_s_gpu", "m_sigma_s2_gpu",
                                                           "m_mix_s_gpu")], rtol=2e-4)


if __name__ == "__main__":
    import sys
    seed_ = None if sys.argv[7] == "None" else sys.argv[7]
    (_pvr_worker if len(sys.argv) > 8 and sys.argv[8] == "pvr" else _worker)(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], bool(int(sys.argv[5])),
                                                                              bool(int(sys.argv[6])), seed_)
