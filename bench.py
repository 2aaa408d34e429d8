#!/usr/bin/env python
"""Headline benchmark: MVoxels/s per SR iteration (PSF forward + back-projection), SVR.

  python bench.py --gpus N --steps K --warmup W

A "step" is one super-resolution iteration of the reference's hot loop (reconstruction.cc:1013-1108, bias correction
off): Scale -> Superresolution (back-projection, all-reduce, Prep + regulariser) -> SimulateSlices (forward) -> MStep ->
EStep, driven by the C++ host object (csrc/svr_host.cpp) on seeded synthetic stacks that are resident in HBM before the
timed region.  value = Va / t_step / 1e6, Va = slice pixels with s != -1 and v_PSF_sums != 0 over all ranks.

Workloads are FIXED (strong scaling, as BASELINE.json's metric and configs name them):
  P4  (default) BASELINE configs[1]: 4 stacks 100x93x70 on the reference's bundled mask geometry, 1.0 mm (workloads.py)
  S8            BASELINE configs[3]: 8 stacks of 64 x 256^2 slices, 0.75 mm
  PVR4          BASELINE configs[2]: 32 x 32 patches, stride 16, of the P4 stacks, 1.0 mm -- the patch-to-volume loop
                (csrc/pvr_host.cpp): Scale -> scatter (+ all-reduce) + regulariser -> simulate -> M-step -> E-step
  PVR8spx       BASELINE configs[4]: superpixel patches (--spxSize 32 --spxExtend 2) of the 8 S8 stacks, 0.5 mm
  P4s / S8h / tiny: the round-1 axis-aligned P4, the S8 stacks at 0.5 mm, the oracle-sized case.
N > 1: one process per GPU, slices (patches) sharded by estimated work, the exchanges on RCCL bound directly by the C library
(csrc/svr_rccl.cpp; `--comm torch` routes them through torch.distributed instead).  Launched by torch.distributed.run --
or by this script itself: `python bench.py --gpus N` without WORLD_SIZE in the environment re-executes under
`python -m torch.distributed.run --nproc-per-node N`.  torch.distributed (gloo) only carries the rendezvous (the 128-byte
ncclUniqueId) and the timing barrier."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this host driver needs dmabuf IPC (RCCL's peer buffers: hipIpcGetMemHandle fails otherwise); the GPU box exports
# it already -- kept here for a launch from a bare environment; must be set before the HIP runtime comes up (before torch is imported)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F32_PEAK_TFLOPS = 157.3      # f32 vector peak with packed FMA on gfx950 (= the f32 MFMA dense peak, same guide)
# flops of one PSF tap (fma = 2).  FORMULA: the reference's per-tap formula, RC.cu:112-130 + 164-174 -- 3 fma (lattice),
# mul + fma (q), sqrt, mul, sin (13), div, exp (18), 3 mul = 49.  EXECUTED: the canonical sequence of round 2 -- 2 fma
# (x', y'), mul + fma (q), mul (h), 3 x (mul, fma, mul), mul (r), rint, sub, mul (s), 4 fma, 2 mul, mul (square), 2 mul
# (Gaussian recurrence), mul = 38, on the taps that are evaluated (dead units cost one tap per row).
FLOPS_PER_TAP_FORMULA = 49
FLOPS_PER_TAP_EXECUTED = 38
TAPS = 4096
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r06_traffic.json")     # written by tools/prof_final.py in the same round: per workload,
                                                                      # per pass: FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU per launch
SIMDS = 1024                 # 256 CUs x 4 SIMDs
VALU_ISSUE_PER_S = 0.6e9     # wave-instructions a SIMD issues per second: 2.4 GHz / 4 cycles per 64-lane instruction
CLOCK_PROBE_REF_MS = 8.50    # svr_clock_probe(2^20) on the boxes of the pool that read 188-189 MVoxels/s on P4 (round 5)
SR_PER_OUTER = 4             # rec_iterations_first (reconstruction.cc:115,187): SR iterations per outer iteration
METRIC = "MVoxels/s per SR iteration (PSF fwd+back), 4-stack 1.0mm SVR, 1/2/4/8 GPU"


def pmc_entry(workload, world, key):
    """What this round's rocprofv3 --pmc passes counted for one pass (key: back / forward / back_table / forward_table) of a
    workload, per launch: {"fetch_bytes", "write_bytes", "valu_insts", "kernels"} or None (file absent, other workload, N > 1)."""
    try:
        tj = json.load(open(TRAFFIC_JSON))
        if world != 1:
            return None
        return tj.get(workload, {}).get(key)
    except Exception:
        return None


def table_traffic(workload, world):
    """FETCH_SIZE / WRITE_SIZE of the table-mode kernels, per launch.  The guide's gfx950 correction applies to these kernels'
    reads: 16 B per lane, coalesced, streaming -- FETCH_SIZE reports half of such bytes -- so `fetch_corrected` doubles it;
    WRITE_SIZE is reported as counted."""
    out = {}
    for key, name in (("back_table", "scatter"), ("forward_table", "gather")):
        e = pmc_entry(workload, world, key)
        if e:
            out[name] = {"fetch_counted": e["fetch_bytes"], "fetch_corrected": 2.0 * e["fetch_bytes"], "write_counted": e["write_bytes"]}
    return out or None


def cpu_baseline(prob):
    """What the line reports as `cpu_baseline`.  SVR workloads: THE REFERENCE'S CPU RECONSTRUCTION PATH -- irtkReconstruction::CoeffInit
    (explicit coefficient lists, once per outer iteration) and one SR iteration = Scale, Superresolution (+ AdaptiveRegularization),
    SimulateSlices, MStep, EStep (reconstruction.cc:1013-1108 with --useCPU, bias correction off; irtkReconstructionGPU.cc:2305-2673,
    1090-1161, 3076-3160, 3442-3695, 3697-3749, 3940-4119, 4121-4263, 4265-4428) -- restated in plain C with pthreads where the reference
    uses TBB (oracle/cpu_twin.c, kind "port": the reference's own sources need GSL / boost / TBB headers this image lacks and are not
    built against stand-ins), on the host cores of this box, on the WHOLE of P4 (every slice; other SVR workloads: every k-th slice so
    that the coefficient lists stay below ~8 GB, said in `sample`).  It is a different algorithm from the GPU path's (Gaussian PSF,
    trilinear splat: README.md:117-119): a reported baseline, not a target and not a parity oracle.  The port of the GPU kernels
    (oracle/svr_oracle.c in literal mode, rounds 1-4's figure) stays beside it as `port_of_gpu_kernels`; patch-based workloads, which
    have no CPU path in the reference, report only that."""
    from fetalreconstruction_amd import engine as _engine
    cores = max(1, min(int(_engine.load_library().svr_host_threads()), 64))     # affinity mask cut to the cgroup CPU quota (16 of 256 on the gpurun boxes)
    port = cpu_baseline_port(prob, target_seconds=6.0)
    if hasattr(prob, "patches_per_stack"):
        return port
    from fetalreconstruction_amd.phantom import sub_problem
    from oracle import cputwin
    ns = prob.ns
    act = int((prob.slices != -1).sum())
    # coefficients per active pixel ~ the transformed PSF's box: (2 d / res + 2)^3 voxels over the three axes, about a third non-zero
    res = float(prob.vdim[0])
    per_px = np.prod([2.0 * float(prob.slice_dim[0][k]) / res + 2.0 for k in range(3)]) / 3.0
    step = max(1, int(np.ceil(act * per_px * 8.0 / 8e9)))
    sub = prob if step == 1 else sub_problem(prob, 0, 0, select=np.arange(0, ns, step))
    tw = cputwin.CpuTwin(sub, threads=cores)
    tw.SetSmoothingParameters(150, 0.02)
    tw.preamble()                          # InitializeEMValues, CoeffInit, GaussianReconstruction, SimulateSlices, InitializeRobustStatistics, EStep
    n_it = 5
    its = []
    for i in range(n_it):
        t0 = time.perf_counter()
        tw.sr_iteration(i % SR_PER_OUTER)
        its.append(time.perf_counter() - t0)
    med = float(np.median(its))
    va = tw.active_pixels
    out = {"value": va / med / 1e6, "unit": "MVoxels/s per SR iteration", "cores": cores, "kind": "port",
           "algorithm": "the reference's CPU reconstruction path (irtkReconstruction::CoeffInit + Scale / Superresolution / SimulateSlices / MStep / EStep, "
                        "irtkReconstructionGPU.cc; Gaussian PSF, trilinear splat, explicit coefficient lists, double) restated in C with pthreads: oracle/cpu_twin.c",
           "coeff_init_s": float(tw.times["CoeffInit"][0]), "sr_iteration_s": med, "sr_iterations_timed": n_it,
           "coefficients": tw.coefficients, "active_pixels": va,
           "sample": (f"the whole workload ({sub.ns} slices, {va} active pixels)" if step == 1 else
                      f"every {step}th slice of the workload ({sub.ns} slices, {va} active pixels: the coefficient lists of all {ns} would need ~{act * per_px * 8 / 1e9:.0f} GB)")
                     + f": CoeffInit once ({tw.times['CoeffInit'][0]:.2f} s, not in `value`), then the median of {n_it} SR iterations ({med:.3f} s) on {cores} threads"
                     + f" = {va / med / 1e6 / cores:.2f} MVoxels/s per thread (BASELINE.md 2: the reference's own CPU twin, compiled against stand-in headers for the survey only, "
                       "read 0.24 MVoxel/s on one thread of the survey's container: the same order -- the port's plausibility check, not a measurement of this box)",
           "per_thread": va / med / 1e6 / cores, "survey_probe_per_thread": 0.24,
           "per_function_s": {k: float(np.mean(v)) for k, v in tw.times.items()},
           "port_of_gpu_kernels": port}
    tw.close()
    return out


def cpu_baseline_port(prob, target_seconds=12.0):
    """The CPU port of the GPU kernels (oracle, literal float32 mode = the reference's own arithmetic) timed on a bounded sample of the same
    workload on the host cores: ONE WHOLE SR ITERATION (Scale, back-projection, Prep + regulariser, forward projection,
    M-step, E-step -- the Python mirror of the host driver on the oracle engine) of every k-th slice.  The sample's
    slices are dealt to one oracle instance per core (its own volume, like the slice-sharded ranks); the C calls
    release the GIL.  A reported baseline, not a target."""
    from concurrent.futures import ThreadPoolExecutor

    from fetalreconstruction_amd.phantom import sub_problem
    from tests.twins.reconstruction import irtkReconstruction
    from oracle import pyoracle as po
    from fetalreconstruction_amd import engine as _engine
    cores = int(_engine.load_library().svr_host_threads())      # affinity mask cut to the cgroup CPU quota (16 of 256 on the gpurun boxes)
    cores = max(1, min(cores, 64))
    act = ((prob.slices > 0) if hasattr(prob, "patches_per_stack") else (prob.slices != -1)).reshape(prob.ns, -1).sum(1)
    per_pixel_s = 2 * 0.14e-3                    # ~0.14 ms / pixel / PSF pass / core on this class of host
    want = max(2000, int(target_seconds / per_pixel_s)) * cores
    step = max(1, int(np.ceil(act.sum() / want)))
    sel = np.arange(0, prob.ns, step)
    cores = min(cores, len(sel))
    parts, load = [[] for _ in range(cores)], np.zeros(cores)
    for i in sel[np.argsort(-act[sel], kind="stable")]:          # heaviest slice first, to the least loaded core
        t = int(np.argmin(load))
        parts[t].append(int(i))
        load[t] += act[i]
    parts = [np.array(sorted(q)) for q in parts if q]
    cores = len(parts)

    is_pvr = hasattr(prob, "patches_per_stack")

    def setup(idx):
        sub = sub_problem(prob, 0, 0, select=idx)
        if is_pvr:                                              # the patch-to-volume loop (pvr.py on the oracle engine)
            from tests.twins import pvr as _pvr
            spx = getattr(prob, "spx_masks", None)
            o = po.OracleReconstruction(sub, po.LITERAL, pvr=True, spx_masks=None if spx is None else np.ascontiguousarray(spx[idx]))
            counts = np.bincount(sub.stack_index, minlength=int(prob.stack_index.max()) + 1)
            d = _pvr.irtkPatchBasedReconstruction(o, counts, prob.min_intensity, prob.max_intensity)
            d.reconstruct_iteration(0)
            d.sr_iteration = lambda i, d=d: _pvr_sr_iteration(d, i)
            return o, d
        o = po.OracleReconstruction(sub, po.LITERAL)
        d = irtkReconstruction(o, sub.ns, max_intensity=prob.max_intensity, min_intensity=prob.min_intensity)
        d.SetSmoothingParameters(150, 0.02)
        d.InitializeEMValuesGPU(); d.GaussianReconstructionGPU(); d.SimulateSlicesGPU()
        d.InitializeRobustStatisticsGPU(); d.EStepGPU()
        return o, d

    def _pvr_sr_iteration(d, i):                                # irtkPatchBasedReconstruction.cpp:505-546
        d.Scale()
        d.e.Superresolution(i + 1, d.patch_weight, d.m_adaptive, float(d.m_alpha), d.m_min_intensity, d.m_max_intensity, float(d.m_delta), float(d.m_lambda))
        d.e.SimulateSlices()
        d.MStep(i + 1)
        d.EStep()

    with ThreadPoolExecutor(cores) as pool:
        pairs = list(pool.map(setup, parts))
        va = int(sum(((o.slices != -1) & (o.psf_sums != 0)).sum() for o, _ in pairs))
        t0 = time.perf_counter()
        list(pool.map(lambda od: od[1].sr_iteration(0), pairs))
        dt = time.perf_counter() - t0
    return {"value": va / dt / 1e6, "unit": "MVoxels/s per SR iteration", "cores": cores, "kind": "port",
            "algorithm": "the GPU path's kernels (reconstruction_cuda2.cu: 16^3 sinc^2 x Gauss taps evaluated in every pass) restated in C: oracle/svr_oracle.c, literal mode",
            "sample": f"every {step}th slice of the workload ({len(sel)} slices, {va} active pixels) dealt to {cores} oracle "
                      f"instances, one thread each: one whole SR iteration (Scale, back-projection, regulariser, forward "
                      f"projection, M-step, E-step) in literal mode in {dt:.1f} s"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12, help="timed steps (a multiple of the schedule's 4 SR iterations per outer iteration: one step in four then rewrites the coefficient table, whatever the warm-up)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="P4", choices=["P4", "P4s", "S8", "S8h", "tiny", "PVR4", "PVR8spx"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-coeff-table", action="store_true", help="skip the second measurement with the coefficient table")
    ap.add_argument("--no-s8", action="store_true", help="skip the S8 record (BASELINE configs[3], measured after the default workload in the same launch)")
    ap.add_argument("--layout", choices=["spatial", "contiguous"], default=None,
                    help="N > 1: how the units are dealt to the ranks (sharding.shard_units): spatial (default) = the r-th part of every stack, contiguous = ranges of the reference's order")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="rccl: the C library's own RCCL collectives; torch: torch.distributed callbacks (--backend)")
    ap.add_argument("--backend", default="nccl", help="--comm torch: torch.distributed backend (nccl = RCCL, gloo)")
    ap.add_argument("--share-gpu", action="store_true", help="dev/test: all ranks use cuda:0 (--comm torch --backend gloo only)")
    ap.add_argument("--force-comm", action="store_true",
                    help="dev/test: run the multi-rank code path (rendezvous, communicator, every exchange of the host loop) at world size 1")
    ap.add_argument("--shard", metavar="r/N | all/N",
                    help="what rank r of an N-rank run costs, measured on THIS one GPU: the workload runs whole to a representative state, then "
                         "rank r's slice / patch range runs alone on a fresh context with that state and the kernels of one SR iteration are timed "
                         "(tools/shard_probe.py).  all/N: every rank in turn + the projected step (kernels + bytes / a stated xGMI rate; labelled a projection)")
    args = ap.parse_args()

    if args.shard:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import shard_probe
        which, n = args.shard.split("/")
        only = None if which == "all" else [int(which)]
        res = shard_probe.run(args.workload, int(n), reps=max(2, args.steps // 2), only=only)
        print(json.dumps(res), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched plainly: become the launcher -- N ranks of this script, one per GPU, on this node
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
        sys.exit(subprocess.call(cmd))

    import torch                                                      # before the engine: one RCCL copy per process
    from fetalreconstruction_amd import engine, phantom, workloads
    from fetalreconstruction_amd.host import RcclComm, irtkPatchBasedReconstruction, irtkReconstruction      # the C++ host objects
    from fetalreconstruction_amd.sharding import TorchComm, patch_cost_weights, shard_units, slice_cost_weights

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dist = None
    multi = world > 1 or args.force_comm
    env_at_start = set(os.environ)
    if args.force_comm:                                               # (the C++ hosts then go through their callbacks at world 1 too)
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")             # one node: the host name need not resolve
        if args.share_gpu:
            local_rank = 0
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, this node shows {torch.cuda.device_count()}: one GPU per rank "
                             f"(--share-gpu runs every rank on GPU 0 for a functional check)")
        torch.cuda.set_device(local_rank)

        def rendezvous():
            if args.comm == "torch" and args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo")                       # rendezvous + barrier only
        own_port = args.force_comm and world == 1 and "MASTER_PORT" not in env_at_start
        for attempt in range(4):
            try:
                rendezvous()
                break
            except Exception as e:                                    # noqa: BLE001 -- DistNetworkError: the port picked above was taken meanwhile
                if not own_port or attempt == 3 or "EADDRINUSE" not in str(e) and "address already in use" not in str(e):
                    raise
                os.environ["MASTER_PORT"] = str(free_port())

    state = {"comm": None, "rccl_world": None}

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    def measure(workload, with_table):
        """one workload through the steps of the contract on this launch's ranks and communicator: W untimed steps, K timed ones between
        barriers -> everything the line reports about it"""
        comm, rccl_world = state["comm"], state["rccl_world"]
        # ---- the workload: fixed, whatever the world size --------------------------------------------------------------
        prob = workloads.get(workload) if workload != "tiny" else phantom.problem_tiny()
        pvr = workload.startswith("PVR")
        if pvr:
            # work of a patch: the pixels that carry data, weighted by orientation (patches of one stack share their geometry)
            work = patch_cost_weights((prob.slices > 0).reshape(prob.ns, -1).sum(1), prob.slice_i2w, prob.slice_t, prob.recon_w2i)
        else:
            act = (prob.slices != -1).reshape(prob.ns, -1).sum(1)
            # work of a slice: estimated PSF work (active pixels x live planes), not the pixel count alone
            work = slice_cost_weights(act, prob.slice_i2w, prob.slice_t, prob.recon_w2i, prob.slice_dim, prob.vdim[0])
        # rank r takes the r-th of `world` work-balanced segments of EVERY stack (sharding.shard_units, layout "spatial": a rank's units
        # are neighbours in space -- about 1 / world of the volume's (cell, plane) items per orientation to stage and to combine instead
        # of all the items of the stacks it holds); `order` = the sharded numbering, which the host object is told
        order, ranges = shard_units(work, prob.stack_index, world, args.layout)
        lo, hi = ranges[rank]
        local = phantom.sub_problem(prob, 0, 0, select=order[lo:hi]) if world > 1 else prob
        spx = getattr(prob, "spx_masks", None)

        rec = engine.Reconstruction(local_rank)
        if pvr:
            rec.set_option("pvr", 1)
            engine.sync_gpu(rec, local, quality_factor=1.0)                # m_quality_factor = 1 (irtkPatchBasedReconstruction.cpp:415)
            if spx is not None:
                rec.set_spx_masks(np.ascontiguousarray(spx[order[lo:hi]]))
        else:
            engine.sync_gpu(rec, local)
        if comm is not None:                              # a second workload in the same launch: the same communicator on the new engine's stream
            if hasattr(comm, "rebind"):
                comm.rebind(rec)
        elif multi:
            if args.comm == "rccl":
                # every rank first proves it can open librccl (a rank that cannot would leave the others waiting inside
                # ncclCommInitRank); if one cannot, all ranks fall back to host-staged exchanges over gloo and say so
                try:
                    uid, why = RcclComm.unique_id(), None
                except Exception as ex:
                    uid, why = None, repr(ex)
                oks = [None] * world
                dist.all_gather_object(oks, why)
                if any(w is not None for w in oks):
                    if rank == 0:
                        print(f"note: RCCL unavailable on some rank ({[w for w in oks if w][0]}); exchanging through gloo", file=sys.stderr)
                    args.comm = "gloo-fallback"
                    comm = TorchComm(device=None)
                else:
                    box = [uid if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    try:
                        comm, why = RcclComm(rec, rank, world, box[0]), None
                    except Exception as ex:               # noqa: BLE001 -- ncclCommInitRank refused (every rank then falls back together)
                        comm, why = None, repr(ex)
                    oks = [None] * world
                    dist.all_gather_object(oks, why)
                    if any(w is not None for w in oks):
                        if rank == 0:
                            print(f"note: the RCCL communicator could not be created ({[w for w in oks if w][0]}); exchanging through gloo", file=sys.stderr)
                        if comm is not None:
                            comm.close()
                        args.comm = "gloo-fallback"
                        comm = TorchComm(device=None)
                    else:
                        rccl_world = comm.rccl_world()
            else:
                comm = TorchComm(device=torch.device("cuda", local_rank) if args.backend == "nccl" else None)
        if multi:
            try:                                          # RCCL's version banner (C stdio, every rank) goes out now, not at exit
                import ctypes                             # after rank 0's JSON line
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
        if pvr:
            drv = irtkPatchBasedReconstruction(rec, prob.patches_per_stack, prob.min_intensity, prob.max_intensity, patch_range=(lo, hi),
                                               comm=comm, force_collectives=args.force_comm)
            if world > 1:
                drv.set_unit_order(order)
            # untimed set-up: the part of an outer iteration before the SR loop (irtkPatchBasedReconstruction.cpp:490-504)
            drv.reconstruct_iteration(0)
        else:
            drv = irtkReconstruction(rec, prob.ns, (lo, hi), comm, prob.max_intensity, prob.min_intensity, force_collectives=args.force_comm)
            if world > 1:
                drv.set_unit_order(order)
            drv.SetSmoothingParameters(150, 0.02)      # reconstruction.cc:99-100 defaults (delta, lambda)
            # untimed set-up: the part of an outer iteration before the SR loop (reconstruction.cc:930-1001)
            drv.InitializeEMValuesGPU()
            drv.GaussianReconstructionGPU()
            drv.SimulateSlicesGPU()
            drv.InitializeRobustStatisticsGPU()
            drv.EStepGPU()
        # the first scatter after new geometry builds its work lists (cell lists, launch order); this untimed one keeps that out of the
        # timed region whatever --warmup is (it only writes addon|cmap and the slice weights, which every SR iteration rebuilds / uploads)
        rec.SuperresolutionBackproject(np.ones(local.ns, np.float32))

        # The steps follow the reference's schedule: an outer iteration re-initialises the EM state and runs rec_iterations_first
        # = 4 SR iterations (reconstruction.cc:115,187,930-1001,1013; irtkPatchBasedReconstruction.cpp:490-504).  Run on without
        # that, the slice-level EM of this workload drops more and more slices (12 of 280 after 2 SR iterations, 117 after 24, 236
        # after 46: tools/check_bench_drift.py) and with them their pixels from the scatter -- a step that gets cheaper the later it
        # is timed.  So every SR_PER_OUTER steps (warm-up and timed alike, by the running step count) the EM part of the outer
        # iteration's preamble runs INSIDE the timed region: InitializeEMValues, InitializeRobustStatistics, EStep -- three small
        # kernels and one host exchange; the volume carries on (registration + Gaussian reconstruction are not part of the metric).
        def em_reinit():
            if pvr:
                drv.initializeEMValues(); drv.InitializeRobustStatistics(); drv.EStep()
            else:
                drv.InitializeEMValuesGPU(); drv.InitializeRobustStatisticsGPU(); drv.EStepGPU()

        done = [0]
        mode_state = {"table": False}

        def step():
            k = done[0] % SR_PER_OUTER
            if done[0] and k == 0:
                em_reinit()
                # an outer iteration brings new slice transformations, and with them new taps: the coefficient table (the default of a
                # slice-to-volume run since round 6) is thrown away HERE, inside the timed region -- the next scatter evaluates and writes
                # the table (coeff_lazy), the other passes of the outer iteration stream it
                if mode_state["table"]:
                    rec.set_option("coeff_invalidate", 1)
            drv.sr_iteration(k)
            done[0] += 1

        def timed(label_steps):
            """K steps between two barriers -> seconds"""
            barrier()
            t_ = time.perf_counter()
            for i in range(label_steps):
                step()
            barrier()
            return time.perf_counter() - t_

        def run_mode(table_on):
            """W untimed steps, then K timed ones with the kernel timers OFF (the figure the line reports), then the same K steps once more
            with HIP events around each hot kernel for `kernel_ms` (the events cost 4-8 us of stream time apiece, a dozen per SR iteration:
            ~1 % of a P4 step on one GPU, several per cent of a rank's share of it) -- the schedule restarted from the same position"""
            if not pvr:
                rec.set_option("coeff_table", 1 if table_on else 0)
            mode_state["table"] = bool(table_on) and not pvr and rec.get_option("coeff_table") == 1
            done[0] = 0
            em_reinit()
            if mode_state["table"]:
                rec.set_option("coeff_invalidate", 1)
            rec.timer_enable(False)
            for i in range(args.warmup):
                step()
            d_ = timed(args.steps)
            fits = pvr or (not table_on) or rec.get_option("coeff_table") == 1     # the table switches itself off when it does not fit the free memory
            rec.timer_enable(True)       # HIP events on the engine's own stream around each hot kernel (svr_timer_*)
            done[0] = 0
            em_reinit()
            if mode_state["table"]:
                rec.set_option("coeff_invalidate", 1)
            for i in range(min(args.warmup, SR_PER_OUTER)):
                step()
            rec.timer_reset()
            d_t = timed(args.steps)
            tm_ = rec.timers()
            rec.timer_enable(False)
            return d_, d_t, tm_, fits

        table_default = (not pvr) and rec.get_option("coeff_table") == 1
        dt, dt_timers, timers, table_fits = run_mode(table_default)
        table_used = table_default and table_fits and rec.get_option("coeff_table") == 1
        cnt = rec.counters()

        def per_rank(tm):
            """what every rank measured on its own engine: average milliseconds per launch of the two PSF kernels, of the volume
            all-reduce (HIP events around the collective on the engine's stream: the wait for the slowest rank + the ring) and of
            the small host-side exchanges (wall clock), plus the rank's share of the work -- so that a scaling shortfall can be
            put down to imbalance, RCCL or the host exchanges from the line alone"""
            keys = ("backproject", "forward", "allreduce", "reduce_scatter", "allgather", "exchange_host", "regularize")
            mine = [tm[k][0] / max(tm[k][1], 1) for k in keys] + [float(tm["exchange_host"][1]) / max(args.steps, 1), float(cnt["Va"]), float(hi - lo)]
            mine.append(float(coll_bytes))
            n = len(mine)
            v = np.zeros(world * n)
            v[rank * n:(rank + 1) * n] = mine
            v = comm.allreduce_sum(v).reshape(world, n) if multi else v.reshape(1, n)
            out = {f"{k}_ms": [round(float(x), 4) for x in v[:, j]] for j, k in enumerate(keys)}
            out["exchanges_per_step"] = [round(float(x), 2) for x in v[:, len(keys)]]
            out["Va"] = [int(x) for x in v[:, len(keys) + 1]]
            out["units"] = [int(x) for x in v[:, len(keys) + 2]]
            # what a rank sends per SR iteration through the volume collectives (slab update: (N-1)/N of reduce-scatter + all-gather
            # messages; replicated update: the ring all-reduce's 2 (N-1)/N of the pair)
            out["collective_bytes_sent"] = [int(x) for x in v[:, len(keys) + 3]]
            return out

        # bytes of the volume collectives per SR iteration and rank
        nvv = float(cnt["Nv"])
        if not multi:
            coll_bytes = 0.0
        elif timers["reduce_scatter"][1]:
            try:
                rsn, agn = rec.slab_chunks(world if world > 1 else 1, rank)
            except Exception:
                rsn, agn = 0, 0
            coll_bytes = 4.0 * (world - 1) * (rsn + agn) if world > 1 else 4.0 * (rsn + agn)
        else:
            coll_bytes = 2.0 * (world - 1) / max(world, 1) * 2.0 * nvv * 4.0
        ranks = per_rank(timers)
        if multi:
            dt = float(comm.allreduce_max(np.array([dt]))[0])
            va = int(round(comm.allreduce_sum(np.array([float(cnt["Va"])]))[0]))
        else:
            va = cnt["Va"]
        tuned = {k: rec.get_option(k) for k in ("fwd_tile_w", "fwd_tile_h", "tile_w", "tile_h", "wave_cap", "back_mode", "fwd_mode", "cell_w", "cell_h", "cell_gw", "cell_gh")}

        # ---- the same K steps in the OTHER mode, reported next to the headline ------------------------------------------------------
        # slice-to-volume: every tap evaluated in every pass (the reference GPU kernels' way, the headline until round 5); patch-based: the
        # coefficient table (not its default: no gain there), written by k_coeff_build outside the timed region as in rounds 2-5
        tab = None
        alt = None
        if with_table:
            try:
                if not pvr:
                    d2, d2t, tm2, _ = run_mode(not table_default)
                    if multi:
                        d2 = float(comm.allreduce_max(np.array([d2]))[0])
                    alt = {"table": not table_default, "dt": d2, "dt_timers": d2t, "timers": tm2}
                    rec.set_option("coeff_table", 1 if table_default else 0)
                else:
                    rec.set_option("coeff_table", 1)
                    rec.timer_enable(True)
                    rec.timer_reset()
                    rec.SimulateSlices()                                  # untimed: builds the table, times the shapes again
                    build_ms = rec.timers()["coeff_build"][0]
                    rec.SuperresolutionBackproject(np.ones(local.ns, np.float32))
                    done[0] = 0                                           # the same schedule from the same EM state as above
                    em_reinit()
                    for i in range(args.warmup):
                        step()
                    on = rec.get_option("coeff_table") == 1               # it switches itself off when it does not fit the free memory
                    rec.timer_reset()
                    dt2 = timed(args.steps)
                    tm2 = rec.timers()
                    rec.timer_enable(False)
                    if multi:
                        dt2 = float(comm.allreduce_max(np.array([dt2]))[0])
                        on = bool(comm.allreduce_min(np.array([1.0 if on else 0.0]))[0] > 0.5)
                    tab = {"on": on, "dt": dt2, "timers": tm2, "bytes": float(cnt["Va"]) * 9 * 1024.0, "build_ms": build_ms}
                    rec.set_option("coeff_table", 0)
            except Exception as ex:                              # the headline line must still be printed
                if multi:
                    raise                                         # (a rank that carried on alone would leave the others in a barrier)
                tab = alt = None
                if rank == 0:
                    print(f"note: the measurement of the other mode failed: {ex!r}", file=sys.stderr)

        state["comm"], state["rccl_world"] = comm, rccl_world
        try:
            uc = rec.unit_counts()
        except Exception:
            uc = None
        # launches that left the cell path (svr_fallbacks): a non-zero entry on a benchmarked workload means float atomics (a scatter) or
        # the slower tile kernels took over without the line's kernel names saying so -- summed over the ranks
        fb = rec.fallbacks()
        # what this box's shader clock does under a full vector load, right after the timed steps (svr_clock_probe): the pool's boxes differ by
        # up to 16 % with one binary, and the line should say which kind of box it was measured on
        try:
            pms, _ = rec.clock_probe()
            probe = {"name": torch.cuda.get_device_name(local_rank), "clock_probe_ms": pms, "clock_probe_reference_ms": CLOCK_PROBE_REF_MS,
                     "relative_clock": CLOCK_PROBE_REF_MS / pms,
                     "note": "2^20 packed f32 fmas per lane (eight independent chains: the vector pipe at its full issue rate) on 4 wavefronts per SIMD of the whole chip, shortest of three "
                             "launches right after the timed steps (rank 0's device): a fixed number of shader cycles, so its time is the inverse of the clock the box sustains under a full vector load.  reference = the "
                             "boxes that read 188-189 MVoxels/s on P4 in round 5; relative_clock well below 1 = a slow box of the pool (DESIGN 7), not a slower binary"}
        except Exception as ex:                                   # noqa: BLE001 -- an aid, never a reason to lose the line
            probe = {"error": repr(ex)}
        if multi:
            fbv = comm.allreduce_sum(np.array([float(fb[k]) for k in sorted(fb)]))
            fb = {k: int(round(x)) for k, x in zip(sorted(fb), fbv)}
        return dict(prob=prob, pvr=pvr, rec=rec, drv=drv, local=local, lo=lo, hi=hi, dt=dt, dt_timers=dt_timers, timers=timers, cnt=cnt, ranks=ranks, va=va, tuned=tuned,
                    tab=tab, alt=alt, table_used=table_used, order=order, unit_counts=uc, cell_order=rec.get_option("cell_order"), fallbacks=fb, device=probe)

    m = measure(args.workload, not args.no_coeff_table)
    prob, pvr, rec, dt, timers, cnt, ranks, va, tuned, tab = (m[k] for k in ("prob", "pvr", "rec", "dt", "timers", "cnt", "ranks", "va", "tuned", "tab"))
    alt, table_used, dt_timers = m["alt"], m["table_used"], m["dt_timers"]
    unit_counts, cell_order, fallbacks, device = m["unit_counts"], m["cell_order"], m["fallbacks"], m["device"]
    comm, rccl_world = state["comm"], state["rccl_world"]
    # ---- BASELINE's multi-GPU target is quoted on S8 (configs[3]: 8 stacks of 64 x 256^2, 0.75 mm), the metric on P4: every line also
    # carries an "s8" record measured in the SAME launch after the headline -- same ranks, same communicator (svr_comm_rebind), same
    # steps and warm-up -- so that a scaling run of the default command answers the S8 question as well.  The headline is unchanged.
    s8 = None
    if not args.no_s8 and args.workload == "P4":
        del m
        rec.close()
        try:
            m8 = measure("S8", not args.no_coeff_table)
            s8 = {"workload": f"S8: {int(m8['prob'].stack_index.max()) + 1} synthetic stacks, volume {tuple(m8['prob'].vsize)}, {m8['prob'].ns} slices of "
                              f"{m8['prob'].slices.shape[2]}x{m8['prob'].slices.shape[1]}, recon {m8['prob'].vdim[0]:.3g} mm (BASELINE configs[3])",
                  "value": m8["va"] / (m8["dt"] / max(args.steps, 1)) / 1e6, "unit": "MVoxels/s", "ms_per_step": m8["dt"] / max(args.steps, 1) * 1e3,
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "Va_total": m8["va"], "Nv": m8["cnt"]["Nv"], "slices": m8["prob"].ns,
                  "ranks": m8["ranks"], "collective_bytes_sent": m8["ranks"]["collective_bytes_sent"],
                  "kernel_ms": {k: (v[0] / max(v[1], 1)) for k, v in m8["timers"].items()},
                  "fallbacks": m8["fallbacks"],
                  "mode": "coefficient table, rewritten every 4 steps inside the timed region" if m8["table_used"] else "every tap evaluated in every pass",
                  "on_the_fly": ({"value": m8["va"] / (m8["alt"]["dt"] / max(args.steps, 1)) / 1e6, "ms_per_step": m8["alt"]["dt"] / max(args.steps, 1) * 1e3}
                                 if (m8["alt"] and not m8["alt"]["table"]) else None),
                  "note": "measured in the same launch after the headline workload, on the same ranks and the same communicator; kernel timers off in the timed steps "
                          "(kernel_ms: the same steps once more with them)"}
            m8["rec"].close()
        except Exception as ex:                                  # the headline line must still be printed
            if multi:
                raise
            s8 = {"error": repr(ex)}

    if rank == 0:
        steps = max(args.steps, 1)
        ms_step = dt / steps * 1e3
        value = va / (dt / steps) / 1e6
        vs, va_l, nv = cnt["Vs"], cnt["Va"], cnt["Nv"]
        # B_back = 4 Vs + 12 Va + 12 Nv, B_fwd = 4 Vs + 13 Va + 8 Nv algorithmic bytes per launch (this rank)
        b_back = 4.0 * vs + 12.0 * va_l + 12.0 * nv
        b_fwd = 4.0 * vs + 13.0 * va_l + 8.0 * nv
        taps = (4096 if not pvr else 1728)
        flops = float(va_l) * taps * FLOPS_PER_TAP_FORMULA
        # taps actually evaluated: every tap of a live (pixel, plane) unit, the first tap of every row of a dead one
        try:
            uc = unit_counts
            nsup = 12 if pvr else 16
            taps_exec = (uc["live_units"] * nsup * nsup + uc["dead_units"] * nsup) / max(uc["pixels"], 1)
            dead_share = uc["dead_units"] / max(uc["live_units"] + uc["dead_units"], 1)
        except Exception:
            taps_exec, dead_share = float(taps), None
        flops_exec = float(va_l) * taps_exec * FLOPS_PER_TAP_EXECUTED

        def entry(kernel, avg, n, alg_bytes, key):
            e = pmc_entry(prob.name, world, key)
            # achieved / frac: the flops the kernel EXECUTES (38 per evaluated tap; dead units one tap per row) -- the conservative figure, the one
            # the round-4 review carried; *_formula: the reference's 49-flop formula credited on all taps of every pixel (rounds 1-4's `frac`)
            ach = flops_exec / avg / 1e12 if n else None
            ach_f = flops / avg / 1e12 if n else None
            out_ = {"kernel": kernel, "bound": "valu_f32", "achieved": ach, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": (ach / F32_PEAK_TFLOPS) if ach else None,
                    "achieved_formula": ach_f, "frac_formula": (ach_f / F32_PEAK_TFLOPS) if ach_f else None,
                    "flops_executed": flops_exec, "frac_executed": (flops_exec / avg / 1e12 / F32_PEAK_TFLOPS) if n else None,
                    "valu_issue_frac": (e["valu_insts"] / (avg * SIMDS * VALU_ISSUE_PER_S)) if (e and n and e.get("valu_insts")) else None,
                    "hbm_achieved_gbs": alg_bytes / avg / 1e9 if n else None, "hbm_peak_gbs": HBM_PEAK_GBS,
                    "hbm_frac": (alg_bytes / avg / 1e9 / HBM_PEAK_GBS) if n else None,
                    "algorithmic_bytes": alg_bytes,
                    "traffic": (e["fetch_bytes"] + e["write_bytes"]) if e else None,
                    "traffic_ratio": ((e["fetch_bytes"] + e["write_bytes"]) / alg_bytes) if e else None,
                    "avg_launch_ms": avg * 1e3, "launches": n}
            return out_

        scatter_name = ("back_cell_kernel + k_cell_combine + k_cell_factors (csrc/svr_cell.inc: cell-owned planes, staged, combined in a fixed "
                        "order, no atomics)" if tuned["back_mode"] == 5 else "back_wave_kernel (wave-owned planes per slice tile, atomic flush)")
        gather_name = ("fwd_cell_kernel + k_cell_gather_finish + k_cell_gfactors (csrc/svr_cell.inc: the gather over the same (cell, plane) items)"
                       if tuned.get("fwd_mode") == 2 else "fwd_unit_kernel (unit-based gather per slice tile)")

        def split(total, part):
            """(ms, launches) of the launches of `total` that are not in `part`"""
            return (total[0] - part[0], total[1] - part[1])

        def avg_s(tm_):
            return tm_[0] / max(tm_[1], 1) * 1e-3

        bt, ft, fs, bs = timers["backproject_table"], timers["forward_table"], timers["forward_store"], timers["backproject_store"]
        bp_eval = split(split(timers["backproject"], bt), bs)
        fw_eval = split(split(timers["forward"], ft), fs)
        e_back = entry(scatter_name + " = SuperresolutionKernel3D_tex, RC.cu:408-522", avg_s(bp_eval), bp_eval[1], b_back, "back")
        e_fwd = entry(gather_name + " = simulateSlicesKernel3D_tex, RC.cu:298-404", avg_s(fw_eval), fw_eval[1], b_fwd, "forward") if fw_eval[1] else None
        eval_note = ("f32 VALU bound (4096 PSF taps per pixel, ~5e3 flop per algorithmic byte; no MFMA: a scatter/gather stencil with a "
                     "sequential epsilon-chain per row has no contraction).  `achieved` / `frac` (= `frac_executed`): the 38 flops of the "
                     "canonical sequence on the taps that are evaluated (dead units: one tap per row) against the packed-f32 vector peak; "
                     "`achieved_formula` / `frac_formula` credit the reference's 49-flop per-tap formula on all taps of a pixel (what rounds 1-4 "
                     "called `frac`).  `valu_issue_frac` = SQ_INSTS_VALU per launch / (launch time x 1024 SIMDs x 0.6e9 wave-instructions/s): the "
                     "share of VALU issue slots used, the figure that explains the time.  hbm_*: SURVEY 8d's algorithmic bytes over the same launch "
                     "time.  `traffic`: FETCH_SIZE + WRITE_SIZE per launch; counters from this round's separate rocprofv3 --pmc passes "
                     "(profiles/r06_traffic.json, tools/prof_final.py), null when that file has no entry for the workload.")
        if table_used and bt[1] and ft[1]:
            # The passes that stream the coefficient table: HBM bound.  Algorithmic bytes of a launch = the 1 KiB of every live (pixel, plane) unit
            # (engine's own count: svr_unit_counts) + SURVEY 8d's compulsory bytes of the pass; `traffic` = what the PMC counters saw.
            tbytes = float(unit_counts["live_units"]) * 1024.0 if unit_counts else float(va_l) * 16 * 1024.0

            def tentry(kernel, tm_, alg, key, extra=""):
                a_ = avg_s(tm_)
                e = pmc_entry(prob.name, world, key)
                tr = (2.0 * e["fetch_bytes"] + e["write_bytes"]) if e else None
                return {"kernel": kernel, "bound": "hbm", "achieved": alg / a_ / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / a_ / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes": alg, "traffic": tr, "traffic_ratio": (tr / alg) if tr else None,
                        "frac_counters": (tr / a_ / 1e9 / HBM_PEAK_GBS) if tr else None,
                        "valu_issue_frac": (e["valu_insts"] / (a_ * SIMDS * VALU_ISSUE_PER_S)) if (e and e.get("valu_insts")) else None,
                        "avg_launch_ms": a_ * 1e3, "launches": tm_[1], "note": extra}
            on_cells = tuned.get("fwd_mode") == 2
            e_bt = tentry("back_cell_kernel<16, false, 1> + k_cell_combine_wave + k_cell_factors (csrc/svr_cell.inc: the cell-owned scatter with the units' taps "
                          "streamed from the coefficient table through a ring of registers) = SuperresolutionKernel3D_tex, RC.cu:408-522, over CoeffInit's "
                          "coefficients (RG.cc:2617-2673)", bt, tbytes + b_back, "back_table")
            e_ft = tentry("fwd_cell_kernel<16, false, 2> (the table's rows HBM -> LDS by LDS-DMA, csrc/svr_cell.inc) or fwd_unit_kernel<.., COEFF> on slice tiles "
                          "(coarse slices: svr_simulate_slices) + k_cell_gather_finish = simulateSlicesKernel3D_tex, RC.cu:298-404" if on_cells else
                          "fwd_unit_kernel<.., COEFF>", ft, tbytes + b_fwd, "forward_table")
            e_fs = tentry("fwd_cell_kernel<16, false, 3>: the gather that evaluates every tap AND writes the table (coeff_lazy) = CoeffInit riding on "
                          "simulateSlicesKernel3D_tex", fs, tbytes + b_fwd, "forward_store",
                          "bytes WRITTEN; VALU-bound by its evaluation and write-bound by its stores at the same time") if fs[1] else None
            e_bs = tentry("back_cell_kernel<16, false, 3>: the scatter that evaluates every tap AND writes the table (coeff_lazy: whichever PSF pass comes first after a "
                          "new slice geometry -- in these steps the scatter, in the reconstruction loop pass 2 of the Gaussian reconstruction) = CoeffInit riding on "
                          "SuperresolutionKernel3D_tex", bs, tbytes + b_back, "back_store",
                          "bytes WRITTEN; VALU-bound by its evaluation and write-bound by its stores at the same time") if bs[1] else None
            step_ms = {"scatter_table": bt[0] / steps, "gather_table": ft[0] / steps, "gather_store": fs[0] / steps, "scatter_store": bs[0] / steps,
                       "scatter_evaluate": bp_eval[0] / steps, "gather_evaluate": fw_eval[0] / steps}
            dom, other = (e_bt, e_ft) if bt[0] >= ft[0] else (e_ft, e_bt)
            roof = dict(dom)
            roof["backproject_table" if dom is e_ft else "forward_table"] = other
            roof["forward_store"] = e_fs
            roof["backproject_store"] = e_bs
            roof["evaluate"] = {"backproject": e_back if bp_eval[1] else None, "forward": e_fwd, "note": eval_note
                                + "  (forward: null when every evaluating gather of the timed steps was the one that writes the table: roofline.forward_store)"}
            roof["ms_per_step_by_kind"] = step_ms
            roof["dead_unit_share"] = dead_share
            roof["note"] = ("The step's dominant kernel by time, measured live (HIP events on the engine's stream, second pass of the same steps): the pass that streams "
                            "the coefficient table -- three of four scatters and all four gathers of an outer iteration; the table is thrown away with every outer "
                            "iteration, inside the timed region, and the scatter that follows evaluates and writes it (`backproject_store`).  HBM bound: "
                            "`achieved` = (1 KiB per live (pixel, plane) unit + SURVEY 8d's bytes of the pass) / launch time against the 8 TB/s peak; `traffic` / "
                            "`frac_counters` = 2 x FETCH_SIZE (the guide's gfx950 correction for 16-byte streaming reads) + WRITE_SIZE from this round's separate "
                            "rocprofv3 --pmc passes (profiles/r06_traffic.json).  `evaluate`: the f32-VALU-bound figures of the evaluating launches, as in rounds 1-5.")
        else:
            dom, other = (e_back, e_fwd) if (e_fwd is None or avg_s(bp_eval) >= avg_s(fw_eval)) else (e_fwd, e_back)
            roof = dict(dom)
            roof["dead_unit_share"] = dead_share
            roof["backproject" if dom is e_fwd else "forward"] = other
            roof["note"] = "The dominant kernel of the step, measured live (HIP events on the engine's stream); the other PSF pass next to it.  " + eval_note
        out = {
            "metric": METRIC,
            "value": value, "unit": "MVoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{prob.name}: {int(prob.stack_index.max()) + 1} synthetic stacks, volume {tuple(prob.vsize)}, "
                                   f"{prob.ns} {'patches' if pvr else 'slices'} of {prob.slices.shape[2]}x{prob.slices.shape[1]}, recon {prob.vdim[0]:.3g} mm"
                                   + (" on the reference's bundled mask geometry (oblique, 300-400 mm off the origin)" if prob.name in ("P4", "PVR4") else "")
                                   + (" -- patch-to-volume loop (BASELINE configs[2]: 32x32 patches, stride 16)" if prob.name == "PVR4" else "")
                                   + (" -- patch-to-volume loop (BASELINE configs[4]: superpixel patches, --spxSize 32 --spxExtend 2)" if prob.name == "PVR8spx" else ""),
                       "Vs": vs, "Va_rank0": va_l, "Va_total": va, "Nv": nv, "slices": prob.ns,
                       "schedule": f"outer iterations of {SR_PER_OUTER} SR iterations (rec_iterations_first, reconstruction.cc:115,187): every "
                                   f"{SR_PER_OUTER} steps InitializeEMValues + InitializeRobustStatistics + EStep run inside the timed region, so that "
                                   "the slices the EM drops do not pile up from step to step (24 steps on: 117 of 280 slices at weight 0)",
                       "parallelism": (f"{'patch' if pvr else 'slice'}-sharded x{world}; volume update by z-slabs: reduce-scatter of addon|cmap at the mask's voxels -> every rank "
                                       f"updates its own planes -> all-gather of the new volume (csrc/svr_slab.inc)" if timers["reduce_scatter"][1] else
                                       f"{'patch' if pvr else 'slice'}-sharded x{world}, 1 in-place all-reduce of addon|cmap (float[2 Nv]) per scatter pass, update replicated")
                                      if world > 1 else "1 GPU",
                       "comm": (args.comm if multi else None), "rccl_world": rccl_world, "device": device,
                       "tuned": {"gather_tile": f"{tuned['fwd_tile_w']}x{tuned['fwd_tile_h']}", "scatter_tile": f"{tuned['tile_w']}x{tuned['tile_h']}",
                                 "scatter_box": tuned["wave_cap"], "back_mode": tuned["back_mode"], "fwd_mode": tuned["fwd_mode"],
                                 "cell": f"{tuned['cell_w']}x{tuned['cell_h']}", "gather_cell": f"{tuned['cell_gw']}x{tuned['cell_gh']}", "pin": os.environ.get("SVR_TILE_PIN"),
                                 "cell_order": cell_order, "fallbacks": fallbacks,
                                 "note": "nothing is picked by timing (fwd_autotune 0): cell sizes and tile shapes follow from the geometry, the (cell, plane) items "
                                         "are launched in order of falling work; runs repeat bit for bit.  The tile shapes apply to pass 1 of the Gaussian "
                                         "reconstruction, the coefficient table's gather and the fallback modes"}},
            "ranks": ranks,
            "s8": s8,
            "roofline": roof,
            "kernel_ms": {k: (v[0] / max(v[1], 1)) for k, v in timers.items()},
            "ms_per_step_with_kernel_timers": dt_timers / steps * 1e3,
            "timing": "value / ms_per_step: K steps between two barriers with the kernel timers OFF; kernel_ms, ranks{} and roofline.avg_launch_ms: the same K steps "
                      "once more with a HIP event pair around each hot kernel (ms_per_step_with_kernel_timers)",
        }
        out["config"]["mode"] = ("coefficient table (svr_set_option coeff_table 1, the default of a slice-to-volume context since round 6): the taps of every live (pixel, plane) "
                                 f"unit kept in HBM, THROWN AWAY every {SR_PER_OUTER} steps inside the timed region (an outer iteration's new slice transformations) and "
                                 "rewritten by the next PSF pass, which evaluates them anyway (coeff_lazy; here the step's scatter) -- irtkReconstruction::CoeffInit's _volcoeffs "
                                 "(irtkReconstructionGPU.cc:2305-2673) on the GPU path; results bit-identical to evaluating every tap in every pass" if table_used else
                                 "every tap evaluated in every pass (the reference GPU kernels' way)")
        if world > 1:
            # the first line from real hardware should explain itself: the step this N was PROJECTED to take from one-GPU per-rank kernel times
            # (tools/shard_curve.py -> profiles/r06_shard_projection.json; slab update, 7 links x 76.8 GB/s x 0.5), next to the one just measured
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", "r06_shard_projection.json")))["curves"][prob.name][str(world)]
                proj = {"source": "profiles/r06_shard_projection.json: per-rank kernel times measured rank after rank on ONE GPU + stated link rates; no collective was run",
                        "default_mode_step_ms": (pj.get("table") or pj.get("on_the_fly"))["slab"]["step_ms"] if table_used else pj["on_the_fly"]["slab"]["step_ms"],
                        "on_the_fly_step_ms": pj["on_the_fly"]["slab"]["step_ms"],
                        "note": "the table's projection holds its steady passes only: the step of an outer iteration that rewrites the table is in the measurement, not in the projection"}
                out["projection"] = proj
                out["speedup_vs_projection"] = {"default_mode": proj["default_mode_step_ms"] / ms_step,
                                                "on_the_fly": (proj["on_the_fly_step_ms"] / (alt["dt"] / steps * 1e3)) if (alt is not None and not alt["table"]) else None}
            except Exception as ex:                                # noqa: BLE001 -- no projection for this workload / N: say so, keep the line
                out["projection"] = {"error": f"no committed projection for {prob.name} at N = {world}: {ex!r}"}
                out["speedup_vs_projection"] = None
        if alt is not None and not alt["table"]:
            out["on_the_fly"] = {"value": va / (alt["dt"] / steps) / 1e6, "unit": "MVoxels/s", "ms_per_step": alt["dt"] / steps * 1e3,
                                 "kernel_ms": {k: (v[0] / max(v[1], 1)) for k, v in alt["timers"].items() if k in ("backproject", "forward", "regularize", "estep", "mstep", "scale")},
                                 "note": "the same K steps with svr_set_option(coeff_table, 0): every tap evaluated in every pass like the reference's GPU kernels -- the headline "
                                         "of rounds 1-5; same results bit for bit"}
        elif alt is not None:
            out["coeff_table"] = {"value": va / (alt["dt"] / steps) / 1e6, "unit": "MVoxels/s", "ms_per_step": alt["dt"] / steps * 1e3}
        # the volume update (Prep + regulariser, RC.cu:1944-1969, 2046-2117): the one HBM-bound kernel of the step; SURVEY 8d: 24 B / voxel
        up_ms, up_n = timers["regularize"]
        if up_n:
            up_avg = up_ms / up_n * 1e-3
            slab = bool(timers["reduce_scatter"][1])
            # the roofline figure only where the timer holds the update kernel alone on the whole volume: in a slab run it also covers the
            # unpack / memset / pack kernels around it and 1 / N of the planes, and whole-volume bytes over it would overstate the rate N-fold
            out["update"] = {"kernel": "k_regul_fused (csrc/svr_regul.inc: Prep + regulariser in one pass, LDS plane ring, float32 rsq weights)"
                                       + (" on this rank's z-slab, with the slab update's unpack and pack kernels (csrc/svr_slab.inc)" if slab else ""),
                             "bound": "hbm", "algorithmic_bytes": None if slab else 24.0 * nv, "avg_launch_ms": up_avg * 1e3,
                             "achieved": None if slab else 24.0 * nv / up_avg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": None if slab else 24.0 * nv / up_avg / 1e9 / HBM_PEAK_GBS,
                             "note": "whole-volume algorithmic bytes (SURVEY 8d: 24 B per voxel) over the launch time; not computed for a slab run (N > 1 or "
                                     "--force-comm): the timer then covers 1 / N of the planes plus the unpack / pack kernels"}
        if tab is not None:
            bp2, fw2 = tab["timers"]["backproject"], tab["timers"]["forward"]
            bp2a, fw2a = bp2[0] / max(bp2[1], 1) * 1e-3, fw2[0] / max(fw2[1], 1) * 1e-3
            out["coeff_table"] = {
                "fits": tab["on"],
                "value": (va / (tab["dt"] / steps) / 1e6) if tab["on"] else None, "unit": "MVoxels/s",
                "ms_per_step": tab["dt"] / steps * 1e3,
                # the table is written once per outer iteration; the reference's defaults run 4 outer iterations of 4 / 4 / 4 / 13 SR
                # iterations (reconstruction.cc:187-188): 4 builds spread over 25 SR iterations
                "build_ms": tab["build_ms"],
                "value_amortised": (va / ((tab["dt"] / steps) + tab["build_ms"] * 1e-3 * 4.0 / 25.0) / 1e6) if tab["on"] else None,
                "kernel_ms": {"backproject": bp2a * 1e3, "forward": fw2a * 1e3},
                "table_bytes_rank0": tab["bytes"],
                "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                             "traffic": table_traffic(prob.name, world),
                             # what the counters saw of the table scatter (FETCH_SIZE with the guide's x 2 for 16 B / lane streaming reads,
                             # + WRITE_SIZE) over the same launch time.  (Rounds 2-4 also printed *_upper figures: the whole table's bytes over
                             # the launch time -- bytes that are not all read: dropped.)
                             "frac_counters": (lambda t_: ((t_["scatter"]["fetch_corrected"] + t_["scatter"]["write_counted"]) / bp2a / 1e9 / HBM_PEAK_GBS)
                                               if (t_ and t_.get("scatter") and bp2[1]) else None)(table_traffic(prob.name, world))},
                "note": "the same K steps with svr_set_option(coeff_table, 1): every live (pixel, plane) unit's 256 taps are "
                        "written once per slice geometry (16 KiB per PSF pixel, outside the timed region like the Gaussian pass "
                        "and the tile lists) and streamed by the scatter and the gather -- CoeffInit's _volcoeffs of the "
                        "reference's CPU path on the GPU path.  Same results (gather bit for bit).  Not the headline: `value` "
                        "above evaluates every tap in every pass like the reference's GPU kernels.  Dead units (about a third on P4) are neither "
                        "stored nor read.",
            }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(prob)
            except Exception as ex:  # the bench line must still be printed
                out["cpu_baseline"] = {"error": repr(ex)}
        try:                                          # whatever RCCL / HIP left in the C stdio buffers goes out first:
            import ctypes                             # the JSON line is the last line of stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if comm is not None and hasattr(comm, "close"):
        barrier()
        comm.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
