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


def _problem():
    from fetalreconstruction_amd import phantom
    return phantom.make_problem(3, (40, 36, 10), 1.1, 2.2, None, 1.0, 15.0, seed=11, orientations=("ax", "cor", "sag"), name="two-rank")


def _worker(rank, world, port, outdir, slabs, slab_update=True):
    import torch                                       # before the engine's library: torch carries its own copy of the HIP runtime
    import torch.distributed as dist
    from fetalreconstruction_amd import engine as E, host, phantom
    from fetalreconstruction_amd.sharding import TorchComm, shard_units, slice_cost_weights
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        P = _problem()
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



def _spawn(world, slabs, slab_update, outdir):
    """two worker PROCESSES of this file (not torch.multiprocessing: importing torch into the pytest process next to the engine's library
    puts two HIP runtimes into one process, which corrupts the heap at exit)"""
    import subprocess
    import sys
    port = _free_port()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(world), str(port), outdir, str(int(slabs)), str(int(slab_update))],
                              cwd=root, env=dict(os.environ, PYTHONPATH=root), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


@pytest.mark.timeout(900)
def test_two_processes_on_one_gpu_through_the_cpp_sharded_host():
    from fetalreconstruction_amd import engine as E, host
    P = _problem()
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
            _spawn(2, slabs, slab_update, d)
            runs[key] = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(2)]
    for key, (r0, r1) in runs.items():
        slabs = key != "no device collectives"
        # both ranks end with the same volume and the same host state
        for k in ("recon", "scale", "sw", "pot", "em"):
            assert np.array_equal(r0[k], r1[k]), (slabs, k)
        assert r0["lohi"][0] == 0 and r0["lohi"][1] == r1["lohi"][0] and r1["lohi"][1] == P.ns
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
    # the slab update against the replicated one: rank-ordered sums either way -> the same bits
    a, b = runs["slab"][0], runs["replicated"][0]
    assert np.array_equal(a["recon"], b["recon"]) and np.array_equal(a["scale"], b["scale"]) and np.array_equal(a["sw"], b["sw"]) and np.array_equal(a["em"], b["em"])


if __name__ == "__main__":
    import sys
    _worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], bool(int(sys.argv[5])), bool(int(sys.argv[6])))
