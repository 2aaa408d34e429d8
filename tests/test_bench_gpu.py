"""bench.py on a GPU box: the one-line contract at N = 1; two ranks sharing the GPU through the slice-sharded C++ host (gloo
carries the collectives there: RCCL refuses two ranks on one device); the C library's RCCL communicator at world size 1 on
the one GPU of the box; and, where two devices are visible, `python bench.py --gpus 2` launching its own ranks over RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError(text[-2000:])


@pytest.mark.gpu
def test_bench_line_and_two_ranks_on_one_gpu():
    # (5 steps: the fifth starts an outer iteration -- the table is thrown away and rewritten inside the timed region)
    one = subprocess.run([sys.executable, "bench.py", "--workload", "tiny", "--steps", "5", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = _last_json(one.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in a, key
    assert a["n_gpus"] == 1 and a["steps"] == 5 and a["value"] > 0 and a["dtype"] == "f32" and a["roofline"]["frac"] > 0
    # round 6: the line's mode is the product's default -- the coefficient table, thrown away and rewritten inside the timed region with every
    # outer iteration -- and its dominant kernel streams the table: HBM bound, algorithmic bytes and the kind of every pass in the line
    assert a["scaling"] == "strong" and a["config"]["mode"].startswith("coefficient table") and a["roofline"]["bound"] == "hbm"
    r = a["roofline"]
    assert r["algorithmic_bytes"] > 0 and "traffic" in r and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-9
    assert r["launches"] > 0 and r["backproject_store"]["launches"] >= 1 and r["backproject_store"]["bound"] == "hbm"
    assert set(r["ms_per_step_by_kind"]) == {"scatter_table", "gather_table", "gather_store", "scatter_store", "scatter_evaluate", "gather_evaluate"}
    assert a["ms_per_step_with_kernel_timers"] > 0 and "timers OFF" in a["timing"]
    # ... and the same steps with every tap evaluated in every pass (the headline of rounds 1-5) next to it
    t = a["on_the_fly"]
    assert t["value"] > 0 and t["ms_per_step"] > 0 and t["kernel_ms"]["backproject"] > 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1",
                          "--comm", "torch", "--backend", "gloo", "--share-gpu", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True,
                         timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    b = _last_json(two.stdout)
    assert b["n_gpus"] == 2 and b["value"] > 0
    # strong scaling: the workload is fixed, both ranks together cover the same active pixels
    assert b["config"]["Va_total"] == a["config"]["Va_total"] and b["config"]["slices"] == a["config"]["slices"]
    assert 0 < b["config"]["Va_rank0"] < b["config"]["Va_total"]
    # every rank's own timers and share of the work travel in the line
    k = b["ranks"]
    assert len(k["Va"]) == 2 and sum(k["Va"]) == b["config"]["Va_total"] and sum(k["units"]) == b["config"]["slices"]
    assert min(k["backproject_ms"]) > 0 and min(k["allreduce_ms"]) > 0 and k["exchanges_per_step"] == [2.0, 2.0]


@pytest.mark.gpu
def test_bench_multi_rank_path_at_world_one():
    """`--force-comm`: the code path of `bench.py --gpus N` -- gloo rendezvous, the ncclUniqueId broadcast, the C library's RCCL
    communicator next to torch's own copy of librccl in one process, every exchange of the sharded host loop -- at world size 1
    on the one GPU of the box.  Same workload, so the same active pixels and (RCCL at world 1 is the identity) a sane rate."""
    one = subprocess.run([sys.executable, "bench.py", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    forced = subprocess.run([sys.executable, "bench.py", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--force-comm"],
                            cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0 and forced.returncode == 0, forced.stderr[-3000:]
    a, b = _last_json(one.stdout), _last_json(forced.stdout)
    assert b["config"]["comm"] == "rccl" and b["config"]["rccl_world"] == 1, b["config"]
    assert b["config"]["Va_total"] == a["config"]["Va_total"] and b["value"] > 0.3 * a["value"]


@pytest.mark.gpu
def test_bench_line_carries_the_s8_record_and_no_fallbacks():
    """The default command line (P4) also measures S8 -- BASELINE configs[3], the workload the >= 6x target is quoted on -- in the same
    launch, on the same ranks and the same communicator (svr_comm_rebind), and reports it as the line's "s8" record next to the unchanged
    headline; here through the forced-collective path at world 1 (reduce-scatter -> slab -> all-gather on RCCL, one host exchange per
    step).  Neither workload may leave the cell path: a fallback (float atomics in the scatter, the tile kernels elsewhere) is counted by
    svr_fallbacks and reported in config.tuned.fallbacks / s8.fallbacks."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-coeff-table", "--force-comm"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    a = _last_json(r.stdout)
    assert a["config"]["workload"].startswith("P4") and a["config"]["comm"] == "rccl" and a["config"]["rccl_world"] == 1
    assert a["config"]["tuned"]["fallbacks"] == dict(scatter_to_atomics=0, gather_to_tiles=0, gauss1_to_tiles=0, tiles_rerun=0)
    s8 = a["s8"]
    assert "error" not in s8, s8
    assert s8["workload"].startswith("S8") and s8["slices"] == 512 and s8["n_gpus"] == 1 and s8["steps"] == 3
    assert 50 < s8["value"] < 400 and abs(s8["value"] - s8["Va_total"] / s8["ms_per_step"] / 1e3) < 1e-6 * s8["value"]
    k = s8["ranks"]
    assert k["Va"] == [s8["Va_total"]] and k["units"] == [512]
    assert k["exchanges_per_step"] == [0.0]                  # no host exchange inside an SR iteration: the slice-level EM runs on the device (csrc/svr_em.inc; steps 1..3: no EM re-initialisation in the timed region)
    assert k["backproject_ms"][0] > 5 and k["forward_ms"][0] > 5 and k["reduce_scatter_ms"][0] > 0 and k["allgather_ms"][0] > 0 and k["allreduce_ms"][0] == 0
    assert s8["collective_bytes_sent"] == k["collective_bytes_sent"] and k["collective_bytes_sent"][0] > 1e7
    assert s8["fallbacks"] == dict(scatter_to_atomics=0, gather_to_tiles=0, gauss1_to_tiles=0, tiles_rerun=0)
    # the headline is the P4 figure, not S8's
    assert a["config"]["Va_total"] < 2e6 < s8["Va_total"] and a["ms_per_step"] < 0.5 * s8["ms_per_step"]


@pytest.mark.gpu
def test_rccl_communicator_of_the_c_library(tiny, oracle_mod):
    """csrc/svr_rccl.cpp on the one GPU of the box: librccl is opened, a communicator of world size 1 is made on the engine's
    stream, and the sharded C++ host runs a whole iteration through its three collectives (identity at world 1) -- the same
    result as without a communicator."""
    import numpy as np
    from fetalreconstruction_amd import engine as E, host
    recs, vols = [], []
    for use_comm in (False, True):
        rec = E.Reconstruction(0)
        E.sync_gpu(rec, tiny)
        comm = None
        if use_comm:
            comm = host.RcclComm(rec, 0, 1, host.RcclComm.unique_id())
            assert comm.rccl_world() == 1
            assert np.array_equal(comm.allreduce_sum(np.array([1.5, 2.0])), [1.5, 2.0])
            assert np.array_equal(comm.allreduce_max(np.array([3.0])), [3.0]) and np.array_equal(comm.allreduce_min(np.array([-1.0])), [-1.0])
        d = host.irtkReconstruction(rec, tiny.ns, (0, tiny.ns), comm, tiny.max_intensity, tiny.min_intensity,
                                    force_collectives=use_comm)          # world 1 through the callbacks (csrc/svr_host.cpp)
        d.SetSmoothingParameters(150, 0.02)
        d.reconstruct_iteration(2)
        vols.append(rec.syncCPU().copy())
        recs.append((rec, comm, d))
    assert np.array_equal(vols[0] == -1, vols[1] == -1)
    assert np.abs(vols[0] - vols[1]).max() <= 2e-5 * np.abs(vols[0]).max()      # float atomics in run-dependent order
    if True:
        # the all-reduce and the host exchanges were really taken and timed (SVR_T_ALLREDUCE / SVR_T_EXCHANGE)
        rec, comm, d = recs[1]
        rec.timer_enable(True)
        rec.timer_reset()
        d.sr_iteration(2)
        tm = rec.timers()
        # (the volume update by z-slabs: one reduce-scatter and one all-gather instead of the all-reduce of the pair)
        # ... and NO host exchange: the M-step's sums (round 4) and the E-step's potentials (round 5: the slice-level EM, csrc/svr_em.inc) meet on the device
        assert tm["allreduce"][1] == 0 and tm["reduce_scatter"][1] == 1 and tm["allgather"][1] == 1 and tm["exchange_host"][1] == 0
        assert tm["reduce_scatter"][0] > 0 and tm["allgather"][0] > 0
    recs[1][1].close()


@pytest.mark.gpu
def test_bench_launches_its_own_rccl_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment spawns two ranks (one GPU each) whose collectives run
    on RCCL, and says which world size RCCL saw.  Needs two visible devices (the gpurun box has one: skipped there)."""
    from fetalreconstruction_amd import engine  # (not torch: a second RCCL copy in this process, after the C library opened its own)
    if engine.device_count() < 2:
        pytest.skip("one visible device")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    one = subprocess.run([sys.executable, "bench.py", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600, env=env)
    two = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert one.returncode == 0 and two.returncode == 0, two.stderr[-3000:]
    a, b = _last_json(one.stdout), _last_json(two.stdout)
    assert b["n_gpus"] == 2 and b["config"]["rccl_world"] == 2 and b["config"]["comm"] == "rccl"
    assert b["config"]["Va_total"] == a["config"]["Va_total"] and 0 < b["config"]["Va_rank0"] < b["config"]["Va_total"]


@pytest.mark.gpu
def test_bench_pvr_workload_through_the_sharded_path():
    """`bench.py --workload PVR4` (BASELINE configs[2]: 32 x 32 patches, stride 16, of the four P4 stacks, cut by the product's own
    command line) through the multi-rank code path at world 1: the line carries the per-rank timers of the two PSF kernels, the
    volume all-reduce and the host exchanges, and the patches' share per rank."""
    r = subprocess.run([sys.executable, "bench.py", "--workload", "PVR4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-coeff-table",
                        "--force-comm"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    a = _last_json(r.stdout)
    assert a["value"] > 0 and a["config"]["comm"] == "rccl" and a["config"]["rccl_world"] == 1 and "patches" in a["config"]["workload"]
    k = a["ranks"]
    assert len(k["Va"]) == 1 and k["Va"][0] == a["config"]["Va_total"] and k["units"][0] == a["config"]["slices"]
    assert k["backproject_ms"][0] > 0 and k["forward_ms"][0] > 0
    assert k["reduce_scatter_ms"][0] > 0 and k["allgather_ms"][0] > 0 and k["collective_bytes_sent"][0] > 0       # the slab update's two collectives
    # no host exchange inside an SR iteration since the patch-level EM runs on the device too (csrc/svr_em.inc, patch form; until then one:
    # the E-step's potentials): the M-step's sums and every rank's potentials / scales meet on the device
    assert k["exchanges_per_step"][0] == 0.0 and k["exchange_host_ms"][0] == 0.0
    assert set(a["config"]["tuned"]) >= {"gather_tile", "scatter_tile", "scatter_box"}


@pytest.mark.gpu
def test_only_the_masks_bounding_box_is_exchanged(tiny):
    """A sharded run all-reduces the mask's bounding box of a volume pair, not the pair (svr_pair_pack / svr_pair_unpack,
    csrc/svr_shard.h): the scatter only ever writes mask voxels.  On the tiny problem the box is 56 % of the volume: pack ->
    (the collective would run on the packed buffer) -> unpack gives the pair back bit for bit, and nothing outside the box is
    non-zero to begin with."""
    import ctypes as C
    import numpy as np
    from fetalreconstruction_amd import engine as E
    rec = E.Reconstruction(0)
    E.sync_gpu(rec, tiny)
    ones = np.ones(tiny.ns, np.float32)
    rec.UpdateScaleVector(ones, ones)
    rec.InitializeEMValues()
    rec.GaussianReconstruction()
    rec.SimulateSlices()
    rec.debug_set(E.BUF_WEIGHTS, np.where(tiny.slices != -1, 0.5, 0).astype(np.float32))
    rec.SuperresolutionBackproject(ones)
    a0, c0 = rec.debug_get(E.BUF_ADDON).copy(), rec.debug_get(E.BUF_CONFIDENCE_MAP).copy()
    vx, vy, vz = tiny.vsize
    m = tiny.mask.reshape(vz, vy, vx) > 0
    nzv = np.argwhere(m)
    lo, hi = nzv.min(0), nzv.max(0)
    box = np.zeros_like(m)
    box[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] = True
    assert (a0.reshape(vz, vy, vx)[~box] == 0).all() and (c0.reshape(vz, vy, vx)[~box] == 0).all() and (c0 > 0).sum() > 1000
    ptr, n = C.c_void_p(), C.c_size_t()
    rec._ck(rec._lib.svr_pair_pack(rec._h, E.BUF_ADDON, C.c_size_t(2 * tiny.nvox), C.byref(ptr), C.byref(n)))
    assert ptr.value and n.value == 2 * int(box.sum()) and n.value < 0.8 * 2 * tiny.nvox
    rec.debug_set(E.BUF_ADDON, np.full(tiny.nvox, 7.0, np.float32))        # (what unpack must overwrite inside the box)
    rec._ck(rec._lib.svr_pair_unpack(rec._h, E.BUF_ADDON, C.c_size_t(2 * tiny.nvox)))
    a1, c1 = rec.debug_get(E.BUF_ADDON), rec.debug_get(E.BUF_CONFIDENCE_MAP)
    assert np.array_equal(a1.reshape(vz, vy, vx)[box], a0.reshape(vz, vy, vx)[box]) and np.array_equal(c1, c0)
    assert (a1.reshape(vz, vy, vx)[~box] == 7.0).all()
    rec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("through_comm", [False, True])
def test_slice_level_em_on_the_device_against_the_host_form(tiny, monkeypatch, through_comm):
    """Round 5 (csrc/svr_em.inc): the host half of EStepGPU -- the two-class EM over the slices, irtkReconstructionGPU.cc:3282-3420 -- as one
    workgroup behind the E-step's kernels, fed by the launcher's all-gather of every rank's potentials / scales / slice_inside: an SR iteration
    makes NO host exchange and waits for nothing.  Against the host form (SVR_DEVICE_SLICE_EM=0) on the same problem: the same excluded
    slices, slice weights within 1e-6 (the five sums over the slices are taken by 256 threads and a tree instead of one after the other:
    ~1e-15 before they are rounded to float), EM scalars to 1e-6 relative, the volume to the float-sum tolerance -- one rank without a
    communicator, and through the C library's RCCL communicator at world 1 (all-gather, reduce-scatter, slab update).  A force-excluded slice
    stays excluded; reading the state in the middle of the loop (svrh_get_state: the device's copy comes over in one wait) changes nothing."""
    import numpy as np
    from fetalreconstruction_amd import engine as E, host
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SVR_DEVICE_SLICE_EM", mode)
        rec = E.Reconstruction(0)
        E.sync_gpu(rec, tiny)
        comm = host.RcclComm(rec, 0, 1, host.RcclComm.unique_id()) if through_comm else None
        d = host.irtkReconstruction(rec, tiny.ns, (0, tiny.ns), comm, tiny.max_intensity, tiny.min_intensity, force_collectives=through_comm)
        d.SetForceExcludedSlices([3])
        d.SetSmoothingParameters(150, 0.02)
        d.reconstruct_iteration(2)
        mid = d.state()                                            # (a read in the middle: pull, then the host's copy goes back up)
        rec.timer_enable(True)
        rec.timer_reset()
        for i in range(2, 4):
            d.sr_iteration(i)
        n_exchanges = rec.timers()["exchange_host"][1]
        st = d.state()
        out[mode] = (rec.syncCPU().copy(), st, mid, n_exchanges)
        if comm:
            comm.close()
        rec.close()
    (v0, s0, m0, x0), (v1, s1, m1, x1) = out["0"], out["1"]
    assert x1 == 0 and x0 == (2 if through_comm else 0)           # host form, sharded path: one exchange per SR iteration
    for a, b in ((m0, m1), (s0, s1)):
        assert np.array_equal(a["slice_weight"] == 0, b["slice_weight"] == 0) and a["slice_weight"][3] == 0 and b["slice_weight"][3] == 0
        assert np.abs(a["slice_weight"] - b["slice_weight"]).max() <= 1e-6
        assert np.allclose(a["scale"], b["scale"], rtol=1e-6) and np.array_equal(a["slice_inside"], b["slice_inside"])
        assert np.allclose(a["slice_potential"], b["slice_potential"], rtol=1e-5, atol=1e-7)
        for k in ("sigma", "mix", "m", "mean_s", "mean_s2", "sigma_s", "sigma_s2", "mix_s"):
            assert a[k] == pytest.approx(b[k], rel=1e-6), k
    assert 0 < (s1["slice_weight"] > 0).sum() < tiny.ns
    assert np.array_equal(v0 == -1, v1 == -1) and np.abs(v0 - v1).max() <= 2e-5 * np.abs(v0).max()


@pytest.mark.gpu
def test_m_step_sums_meeting_on_the_device_give_the_host_exchanges_bits(tiny, monkeypatch):
    """A sharded SR iteration makes ONE host exchange since round 4: the M-step's five sums of every rank are all-gathered on the device,
    added up there in rank order and fed to the E-step (svr_mstep_partial / svr_mstep_estep_ranks, csrc/svr_host.cpp EStepGPU) instead of
    travelling through the hosts (SVR_DEVICE_EM=0: the round-3 form).  Same operations in the same order: the volume and the EM state are
    the same bits.  World 1 through the C library's RCCL communicator (the gpurun box has one GPU)."""
    import numpy as np
    from fetalreconstruction_amd import engine as E, host
    out = {}
    monkeypatch.setenv("SVR_DEVICE_SLICE_EM", "0")        # (the slice-level EM on the host: what this test compares are the two ways of the M-step's sums)
    for mode in ("0", "1"):
        monkeypatch.setenv("SVR_DEVICE_EM", mode)
        rec = E.Reconstruction(0)
        E.sync_gpu(rec, tiny)
        comm = host.RcclComm(rec, 0, 1, host.RcclComm.unique_id())
        d = host.irtkReconstruction(rec, tiny.ns, (0, tiny.ns), comm, tiny.max_intensity, tiny.min_intensity, force_collectives=True)
        d.SetSmoothingParameters(150, 0.02)
        d.reconstruct_iteration(3)
        rec.timer_enable(True)
        rec.timer_reset()
        d.sr_iteration(3)
        n_exchanges = rec.timers()["exchange_host"][1]
        st = d.state()
        out[mode] = (rec.syncCPU().copy(), st["scale"].copy(), st["slice_weight"].copy(), (st["sigma"], st["mix"], st["m"]), n_exchanges)
        comm.close()
        rec.close()
    assert out["0"][4] == 2 and out["1"][4] == 1
    for a, b in zip(out["0"][:3], out["1"][:3]):
        assert np.array_equal(a, b)
    assert out["0"][3] == out["1"][3]


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_the_scale_step_s_command_shape_at_four_ranks_on_one_gpu():
    """The exact command the driver's SCALE step runs -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- at N = 4 with the default workload (P4, then the s8 record in the same launch), on
    the one GPU of this box: `--share-gpu --comm torch --backend gloo` (RCCL refuses several ranks on one device; gloo carries the collectives,
    host-staged).  No hardware claim -- what is asserted is the line: every rank's record, the s8 record's, the bytes the volume collectives
    send, the communicator fields, the projection the line is compared with, and that four ranks cover the fixed workload."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1", "--comm", "torch", "--backend", "gloo",
                        "--share-gpu", "--no-cpu-baseline", "--no-coeff-table"], cwd=ROOT, capture_output=True, text=True, timeout=1400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    a = _last_json(r.stdout)
    assert a["n_gpus"] == 4 and a["steps"] == 2 and a["scaling"] == "strong" and a["value"] > 0 and a["config"]["workload"].startswith("P4")
    assert a["config"]["comm"] == "torch" and a["config"]["rccl_world"] is None and "sharded x4" in a["config"]["parallelism"]
    k = a["ranks"]
    assert len(k["Va"]) == 4 and sum(k["Va"]) == a["config"]["Va_total"] and sum(k["units"]) == a["config"]["slices"] == 280
    assert min(k["backproject_ms"]) > 0 and min(k["forward_ms"]) > 0 and len(k["collective_bytes_sent"]) == 4 and min(k["collective_bytes_sent"]) > 0
    assert max(k["Va"]) < 0.4 * a["config"]["Va_total"]                  # work-balanced shares of a fixed workload
    s8 = a["s8"]
    assert "error" not in s8, s8
    assert s8["n_gpus"] == 4 and len(s8["ranks"]["Va"]) == 4 and sum(s8["ranks"]["Va"]) == s8["Va_total"] and sum(s8["ranks"]["units"]) == 512
    assert s8["collective_bytes_sent"] == s8["ranks"]["collective_bytes_sent"] and min(s8["collective_bytes_sent"]) > 1e6
    assert a["config"]["tuned"]["fallbacks"] == dict(scatter_to_atomics=0, gather_to_tiles=0, gauss1_to_tiles=0, tiles_rerun=0)
    # the projection this N was given (profiles/r06_shard_projection.json) travels with the measurement
    assert "projection" in a and ("default_mode_step_ms" in a["projection"] or "error" in a["projection"])
    if "error" not in a["projection"]:
        assert a["speedup_vs_projection"]["default_mode"] > 0
