#!/usr/bin/env python
"""Headline benchmark: MVoxels/s per SR iteration (PSF forward + back-projection), SVR.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one super-resolution iteration of the reference's hot loop
(reconstruction.cc:1013-1108, bias correction off): Scale -> Superresolution (back-projection,
all-reduce, Prep + regulariser) -> SimulateSlices (forward) -> MStep -> EStep, on seeded synthetic
stacks that are resident in HBM before the timed region.  value = Va / t_step / 1e6 where Va is
the number of slice pixels with s != -1 and v_PSF_sums != 0 summed over all ranks.
Workload at N=1: P4 (4 stacks 100x93x70 of 1.176x1.176x1.25 mm voxels, thickness 2.5 mm,
1.0 mm reconstruction) = BASELINE.json configs[1].  Weak scaling for N>1: every rank gets its own
4 stacks of that shape (n_stacks = 4N), slices sharded by active-pixel count.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PMC_TRAFFIC_BACK_P4 = (110879.0 + 3789632.0) * 1024.0   # bytes per back_plane_kernel launch, see roofline.traffic
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F32_PEAK_TFLOPS = 157.3      # f32 MFMA dense peak == f32 vector peak on gfx950 (same guide)
# algorithmic flops of one PSF tap of the canonical float32 sequence:
# 3 fma (lattice) + q (mul,fma) + sqrt + mul + sin (13) + div + exp (18) + 3 mul = 49 counted as
# fma=2, everything else 1
FLOPS_PER_TAP = 49
TAPS = 4096


def cpu_baseline(prob, target_seconds=10.0):
    """The CPU port (oracle, literal float32 mode) timed on a bounded sample of the same workload on all host
    cores: forward + back-projection of every k-th slice (the rest of an SR iteration is <1 % of the CPU time).
    The sample's slices are dealt to one oracle instance per core (its own partial volume, like the slice-sharded
    GPU ranks); the C calls release the GIL, so the instances run concurrently.  Reported baseline only."""
    from concurrent.futures import ThreadPoolExecutor

    from fetalreconstruction_amd.phantom import sub_problem
    from oracle import pyoracle as po
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    act = (prob.slices != -1).reshape(prob.ns, -1).sum(1)
    per_pixel_s = 2 * 0.14e-3                    # measured ~0.14 ms / pixel / pass / core on this class of host
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

    def setup(idx):
        o = po.OracleReconstruction(sub_problem(prob, 0, 0, select=idx), po.LITERAL)
        o.InitializeEMValues()
        o.GaussianReconstruction()
        o.SimulateSlices()
        return o

    def step_fn(o):
        o.SuperresolutionBackproject(np.ones(o.prob.ns, np.float32))
        o.SimulateSlices()

    with ThreadPoolExecutor(cores) as pool:
        orcs = list(pool.map(setup, parts))
        va = int(sum(((o.slices != -1) & (o.psf_sums != 0)).sum() for o in orcs))
        t0 = time.perf_counter()
        list(pool.map(step_fn, orcs))
        dt = time.perf_counter() - t0
    return {"value": va / dt / 1e6, "unit": "MVoxels/s per SR iteration", "cores": cores, "kind": "port",
            "sample": f"every {step}th slice of the workload ({len(sel)} slices, {va} active pixels) dealt to {cores} oracle "
                      f"instances, one thread each: literal-mode back-projection + forward projection in {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="P4", choices=["P4", "S8", "S8h", "tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="dev/test: all ranks use cuda:0 (gloo only)")
    args = ap.parse_args()

    import torch
    from fetalreconstruction_amd import engine, phantom
    from fetalreconstruction_amd.host import irtkReconstruction          # the C++ host object
    from fetalreconstruction_amd.reconstruction import LocalComm, TorchComm, shard_slices

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        comm = TorchComm(device=torch.device("cuda", local_rank))
    else:
        comm = LocalComm()
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    # weak scaling: 4 stacks per rank of the named shape
    if args.workload == "P4":
        prob = phantom.make_problem(4 * world, (100, 93, 70), 1.17647, 1.25, 2.5, 1.0, 50.0, name="P4")
    elif args.workload == "S8h":      # the grid of BASELINE.json configs[4]: the S8 stacks reconstructed at 0.5 mm (406^3 voxels)
        prob = phantom.make_problem(8 * world, (256, 256, 64), 1.0, 2.5, 2.5, 0.5, 100.0,
                                    orientations=("ax", "cor", "sag"), name="S8h")
    elif args.workload == "S8":
        prob = phantom.make_problem(8 * world, (256, 256, 64), 1.0, 2.5, 2.5, 0.75, 100.0,
                                    orientations=("ax", "cor", "sag"), name="S8")
    else:
        prob = phantom.problem_tiny()
    act = (prob.slices != -1).reshape(prob.ns, -1).sum(1)
    lo, hi = shard_slices(act, world)[rank]
    local = phantom.sub_problem(prob, lo, hi) if world > 1 else prob

    rec = engine.Reconstruction(local_rank)
    engine.sync_gpu(rec, local)
    drv = irtkReconstruction(rec, prob.ns, (lo, hi), comm if world > 1 else None, prob.max_intensity,
                             prob.min_intensity)
    drv.SetSmoothingParameters(150, 0.02)      # reconstruction.cc:99-100 defaults (delta, lambda)

    # untimed set-up: the part of an outer iteration before the SR loop (reconstruction.cc:930-1001)
    drv.InitializeEMValuesGPU()
    drv.GaussianReconstructionGPU()
    drv.SimulateSlicesGPU()
    drv.InitializeRobustStatisticsGPU()
    drv.EStepGPU()
    # the engine times its tile shapes on the first gather / scatter after new geometry ("Tile shapes follow the
    # problem"); the gather was tuned by SimulateSlicesGPU above, this untimed scatter keeps the other one out of the timed
    # region whatever --warmup is (it only writes addon/cmap, which every SR iteration rebuilds, and the slice weights, which
    # every SR iteration uploads)
    rec.SuperresolutionBackproject(np.ones(local.ns, np.float32))
    for i in range(args.warmup):
        drv.sr_iteration(i)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    rec.timer_enable(True)
    rec.timer_reset()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        drv.sr_iteration(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    timers = rec.timers()
    cnt = rec.counters()

    if world > 1:
        dt = float(comm.allreduce_max(np.array([dt]))[0])
        va = int(round(comm.allreduce_sum(np.array([float(cnt["Va"])]))[0]))
    else:
        va = cnt["Va"]

    if rank == 0:
        ms_step = dt / max(args.steps, 1) * 1e3
        value = va / (dt / max(args.steps, 1)) / 1e6
        bp_ms, bp_n = timers["backproject"]
        fw_ms, fw_n = timers["forward"]
        bp_avg = bp_ms / max(bp_n, 1) * 1e-3
        vs, va_l, nv = cnt["Vs"], cnt["Va"], cnt["Nv"]
        # B_back = 4*Vs + 12*Va + 12*Nv algorithmic bytes per launch
        b_back = 4.0 * vs + 12.0 * va_l + 12.0 * nv
        flops = float(va_l) * TAPS * FLOPS_PER_TAP
        out = {
            "metric": "MVoxels/s per SR iteration (PSF fwd+back), 4-stack 1.0mm SVR, 1/2/4/8 GPU",
            "value": value, "unit": "MVoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{prob.name}: {int(prob.stack_index.max()) + 1} synthetic stacks, volume {prob.vsize}, "
                                   f"{prob.ns} slices of {prob.slices.shape[2]}x{prob.slices.shape[1]}, "
                                   f"recon {prob.vdim[0]} mm",
                       "Vs": vs, "Va_rank0": va_l, "Va_total": va, "Nv": nv, "slices": prob.ns,
                       "parallelism": f"slice-sharded x{world}, 1 volume all-reduce per scatter pass"},
            "roofline": {
                "kernel": "back_plane_kernel<8> (SuperresolutionKernel3D_tex, RC.cu:408-522)",
                "bound": "mfma", "achieved": flops / bp_avg / 1e12 if bp_n else None, "peak": F32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": (flops / bp_avg / 1e12 / F32_PEAK_TFLOPS) if bp_n else None,
                # bytes per launch through the L2's memory side on P4 (FETCH_SIZE + WRITE_SIZE, separate
                # rocprofv3 --pmc passes, KB -> B; profiles/r01_f_pmc_back_fwd.txt).  Not measured live.
                "traffic": PMC_TRAFFIC_BACK_P4 if (prob.name == "P4" and world == 1) else None,
                "note": "f32 VALU bound: 4096 PSF taps per pixel, ~5e3 flop per algorithmic byte; peak = dense f32 "
                        "MFMA peak = f32 vector FMA peak on gfx950; `achieved` counts the 49 algorithmic flops per "
                        "tap, the kernel issues ~47 VALU instructions per tap and keeps the VALU 70 % active "
                        "(forward gather: 38 per tap; SQ_ACTIVE_INST_VALU, profiles/r01_f_pmc_back_fwd.txt); "
                        "the HBM view follows",
                "hbm": {"bound": "hbm", "achieved": b_back / bp_avg / 1e9 if bp_n else None, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": (b_back / bp_avg / 1e9 / HBM_PEAK_GBS) if bp_n else None,
                        "algorithmic_bytes": b_back},
                "avg_launch_ms": bp_avg * 1e3, "launches": bp_n,
            },
            "kernel_ms": {k: (v[0] / max(v[1], 1)) for k, v in timers.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(prob)
            except Exception as ex:  # the bench line must still be printed
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
