"""What one rank of an N-rank run costs, measured on ONE GPU (there is no second one on the gpurun box).

The whole workload first runs on one context to a representative state (the reference's preamble + two SR iterations: EM weights,
scales, slice weights, simulated slices, the volume).  Then, for every rank r of N, a fresh context takes rank r's slice / patch
range (the same ranges bench.py --gpus N would use), the donor's state for those units and the donor's volume, and the kernels of
one SR iteration are timed there with HIP events (svr_timer_*): the scatter incl. its combine, the volume update (whole volume =
what a replicated run pays, and the rank's z-slab = what reduce-scatter -> slab -> all-gather pays), the gather, the EM kernels.
Nothing here is a scaling measurement: no collective runs; the projection adds bytes / a STATED link rate and is labelled so.

usage: python tools/shard_probe.py WORKLOAD N [--reps K] [--out FILE]    (also reached as bench.py --shard all/N)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

# xGMI on MI355X: 7 links per GPU, ~153 GB/s each bidirectional = 76.8 GB/s per direction (MI355X_MICROARCH.md).  A reduce-scatter
# or all-gather over a fully connected node moves (N - 1) / N of the message through the N - 1 links of a GPU side by side; ring
# algorithms reach less.  The projection assumes the direct exchange at XGMI_EFF of the wire rate -- an assumption, stated in the output.
XGMI_LINK_GBS = 76.8
XGMI_EFF = 0.5
HOST_EXCHANGE_MS = 0.08      # one stream synchronisation + one small collective (60-100 us measured at world 1)
# per SR iteration.  Both host objects since round 5: none -- the slice- / patch-level EM runs on the device (csrc/svr_em.inc; round 4: one,
# the E-step's potentials); what it costs instead are two SMALL device collectives on the engine's stream (the
# M-step's 16 floats per rank, the potentials' 3 x maxn floats per rank), priced at a stated latency each
HOST_EXCHANGES = {"svr": 0, "pvr": 0}      # (both host objects run their slice- / patch-level EM on the device)
SMALL_COLLECTIVES = {"svr": 2, "pvr": 2}
SMALL_COLLECTIVE_MS = 0.02   # an assumption (RCCL's small-message latency on one node); stated in the output
# the device-side slice- / patch-level EM behind the E-step (k_mstep_scalars_dev + k_slice_em_pack + k_slice_em: one workgroup, latency bound;
# 3.6 + 3.5 + 13-17 us in the kernel traces of a P4 step, the same at S8's 511 slices): the probe times the E-step's kernels through the host
# form, so this is added to every rank's step as a constant
DEVICE_EM_MS = 0.025


def build(wl):
    from fetalreconstruction_amd import workloads, phantom
    return workloads.get(wl) if wl != "tiny" else phantom.problem_tiny()


def make_engine(prob, pvr, spx=None, opts=()):
    from fetalreconstruction_amd import engine
    rec = engine.Reconstruction(0)
    if pvr:
        rec.set_option("pvr", 1)
        engine.sync_gpu(rec, prob, quality_factor=1.0)
        if spx is not None:
            rec.set_spx_masks(np.ascontiguousarray(spx))
    else:
        engine.sync_gpu(rec, prob)
    for k, v in opts:
        rec.set_option(k, v)
    return rec


def kernel_ms(rec, keys=("backproject", "forward", "regularize", "estep", "mstep", "scale")):
    t = rec.timers()
    return {k: (t[k][0] / t[k][1] if t[k][1] else 0.0) for k in keys}


def collective_ms(bytes_per_rank_out, world):
    """direct exchange: a rank sends (N - 1) / N of its message over its N - 1 links side by side"""
    if world <= 1:
        return 0.0
    links = min(world - 1, 7)
    return bytes_per_rank_out * (world - 1) / world / (links * XGMI_LINK_GBS * 1e9 * XGMI_EFF) * 1e3


def probe(wl, world, reps=6, opts=(), only=None, layout=None):
    from fetalreconstruction_amd import engine as E, phantom, host
    from fetalreconstruction_amd.sharding import DEFAULT_LAYOUT, patch_cost_weights, shard_units, slice_cost_weights
    layout = layout or os.environ.get("SVR_SHARD_LAYOUT", DEFAULT_LAYOUT)
    if not os.environ.get("SHARD_HOST_RESTORE"):
        import torch                                   # before the engine's library brings the HIP runtime up: torch carries its own copy
        torch.cuda.init()
    prob = build(wl)
    pvr = wl.startswith("PVR")
    spx = getattr(prob, "spx_masks", None)
    if pvr:
        work = patch_cost_weights((prob.slices > 0).reshape(prob.ns, -1).sum(1), prob.slice_i2w, prob.slice_t, prob.recon_w2i)
    else:
        act = (prob.slices != -1).reshape(prob.ns, -1).sum(1)
        work = slice_cost_weights(act, prob.slice_i2w, prob.slice_t, prob.recon_w2i, prob.slice_dim, prob.vdim[0])
    order, ranges = shard_units(work, prob.stack_index, world, layout)     # the numbering and ranges bench.py --gpus N uses

    # ---- the donor: the whole workload on one context ------------------------------------------------------------------------
    rec = make_engine(prob, pvr, spx, opts)
    if pvr:
        d = host.irtkPatchBasedReconstruction(rec, prob.patches_per_stack, prob.min_intensity, prob.max_intensity)
        d.reconstruct_iteration(0)
        upd = (False, 0.5, float(prob.min_intensity), float(prob.max_intensity), 1.0, 0.1)      # patchBasedSuperresolution_gpu.cu:293-295
    else:
        d = host.irtkReconstruction(rec, prob.ns, max_intensity=prob.max_intensity, min_intensity=prob.min_intensity)
        d.SetSmoothingParameters(150, 0.02)
        d.InitializeEMValuesGPU(); d.GaussianReconstructionGPU(); d.SimulateSlicesGPU(); d.InitializeRobustStatisticsGPU(); d.EStepGPU()
        upd = (False, 0.8, float(prob.min_intensity), float(prob.max_intensity), 150.0, 0.02 * 150.0 ** 2)
    for i in range(2):
        d.sr_iteration(i)
    st = d.state()
    sw = np.asarray(st["patch_weight" if pvr else "slice_weight"], np.float32)
    scales = np.asarray(st["scale"], np.float32)
    donor = {b: rec.debug_get(b) for b in (E.BUF_WEIGHTS, E.BUF_SIMSLICES, E.BUF_SIMWEIGHTS, E.BUF_PSF_SUMS, E.BUF_SIMINSIDE)}
    vol = rec.syncCPU().copy()
    cnt = rec.counters()
    n2 = prob.slices.shape[1] * prob.slices.shape[2]

    em3 = tuple(float(st[k]) for k in (("m_m_gpu", "m_sigma_gpu", "m_mix_gpu") if pvr else ("m", "sigma", "mix")))

    def time_kernels(rec_, sw_, sc_, w_):
        # The state a repetition starts from is put back DEVICE TO DEVICE (saved copies of the weights and the volume; round 5).  Rounds 3-4
        # uploaded them from the host between repetitions: 5-100 ms in which the device sat idle, so that every timed kernel of a rank -- a
        # 5 ms launch -- started on a chip coming out of idle, where a rank of a real run launches its kernels back to back (the first
        # launch after a gap is 5-15 % slower than the following ones: profiles/r05_shard_probe_notes.txt).  SHARD_HOST_RESTORE=1: the old way.
        fast = not os.environ.get("SHARD_HOST_RESTORE")
        if fast:
            import torch
            from fetalreconstruction_amd.sharding import _device_view
            n_w = int(np.asarray(w_).size)
            dev = torch.device("cuda", 0)
            w_saved = torch.from_numpy(np.ascontiguousarray(w_, np.float32).reshape(-1)).to(dev)
            v_saved = torch.from_numpy(np.ascontiguousarray(vol, np.float32).reshape(-1)).to(dev)
            torch.cuda.synchronize()
        rec_.timer_enable(True)
        for k in range(reps + 1):
            if k == 1:
                rec_.timer_reset()
            if os.environ.get("SHARD_WARM"):
                # the restores below leave the device idle for milliseconds and its clocks fall; a rank of a real run launches its kernels
                # back to back -- an untimed scatter first, so that the timed one starts on a busy device
                rec_.timer_enable(False)
                rec_.SuperresolutionBackproject(sw_)
                rec_.timer_enable(True)
            rec_.SuperresolutionBackproject(sw_)
            rec_.SuperresolutionUpdate(*upd)
            rec_.SimulateSlices()
            rec_.MStepSums()                                       # the EM kernels of the step: M-step sums, E-step, scale
            rec_.EStep(*em3)
            rec_.CalculateScaleVector()
            rec_.UpdateScaleVector(sc_, sw_)                       # (the state the scatter started from)
            if fast:
                rec_.stream_sync()
                _device_view(torch, rec_.device_ptr(E.BUF_WEIGHTS), n_w, dev).copy_(w_saved)
                _device_view(torch, rec_.device_ptr(E.BUF_RECONSTRUCTED), int(v_saved.numel()), dev).copy_(v_saved)   # (the volume buffers flip: ask every time)
                torch.cuda.synchronize()
            else:
                rec_.debug_set(E.BUF_WEIGHTS, w_)
                rec_.UpdateReconstructed(rec_.vsize, vol)          # (untimed: every repetition updates the donor's volume)
        out = kernel_ms(rec_)
        rec_.timer_enable(False)
        return out

    full = time_kernels(rec, sw, scales, donor[E.BUF_WEIGHTS])
    full["cells"] = rec.cell_stats()
    full["Va"] = cnt["Va"]
    nv = cnt["Nv"]
    vsize = [int(v) for v in rec.vsize]
    rec.close()

    # ---- rank by rank --------------------------------------------------------------------------------------------------------
    shards = []
    for r, (lo, hi) in enumerate(ranges):
        if only is not None and r not in only:
            continue
        idx = order[lo:hi]
        sub = phantom.sub_problem(prob, 0, 0, select=idx)
        rs = make_engine(sub, pvr, None if spx is None else spx[idx], opts)
        rs.UpdateScaleVector(scales[idx], sw[idx])
        for b, a in donor.items():
            rs.debug_set(b, np.ascontiguousarray(a.reshape(prob.ns, -1)[idx]).reshape(-1))
        rs.UpdateReconstructed(rs.vsize, vol)
        k = time_kernels(rs, np.ascontiguousarray(sw[idx]), np.ascontiguousarray(scales[idx]),
                         np.ascontiguousarray(donor[E.BUF_WEIGHTS].reshape(prob.ns, -1)[idx]).reshape(-1))
        k.update(rank=r, units=[int(lo), int(hi)], Va=rs.counters()["Va"], cells=rs.cell_stats())
        shards.append(k)
        rs.close()
    # what the slab update's collectives carry (svr_slab_plan on a fresh context would need the mask only; the donor's is gone)
    mask_fraction = float((np.asarray(prob.mask) != 0).mean())
    return dict(workload=wl, world=world, layout=layout, Nv=nv, volume=vsize, mask_fraction=mask_fraction, full=full, shards=shards)


def project_at(res, gbs_per_direction):
    """the projection with the collectives priced at ONE stated rate (GB/s a rank can send, all links together): speed-ups slab / replicated"""
    W, nv = res["world"], res["Nv"]
    full, sh = res["full"], res["shards"]
    em = lambda k: k["estep"] + k["mstep"] + k["scale"]
    one = full["backproject"] + full["regularize"] + full["forward"] + em(full) + DEVICE_EM_MS   # (one GPU runs the device EM too)
    psf = max(k["backproject"] + k["forward"] + em(k) for k in sh)
    reg_full = max(k["regularize"] for k in sh)
    ms = lambda nbytes: nbytes * (W - 1) / W / (gbs_per_direction * 1e9) * 1e3 if W > 1 else 0.0
    kind = "pvr" if str(res.get("workload", "")).startswith("PVR") else "svr"
    small = HOST_EXCHANGES[kind] * HOST_EXCHANGE_MS + SMALL_COLLECTIVES[kind] * SMALL_COLLECTIVE_MS + DEVICE_EM_MS
    mfrac = res.get("mask_fraction", 1.0)
    rs_ms, ag_ms, ar_ms = ms(2 * nv * 4 * mfrac), ms(nv * 4 * min(1.0, mfrac * 1.15)), 2 * ms(2 * nv * 4)
    slab = psf + rs_ms + reg_full / W + ag_ms + small
    repl = psf + ar_ms + reg_full + small
    return dict(rate_GBs=gbs_per_direction, collectives_slab_ms=rs_ms + ag_ms, step_slab_ms=slab, speedup_slab=one / slab, allreduce_ms=ar_ms,
                step_replicated_ms=repl, speedup_replicated=one / repl)


# what a rank can send per direction, three ways to look at MI355X's 7 xGMI links of 76.8 GB/s per direction (153.6 bidirectional):
# all 7 side by side at 50 % (a direct exchange: the figure of rounds 3-4), one link's bidirectional figure as a per-direction budget
# (SURVEY 5's ring estimate), and one link in one direction (a ring that keeps a single link busy)
LINK_RATES_GBS = {"7 links x 76.8 GB/s x 0.5 (direct exchange)": 7 * 76.8 * 0.5, "153 GB/s (SURVEY 5: ring over one link pair)": 153.0,
                  "76.8 GB/s (one link, one direction)": 76.8}


def project(res):
    """projected step of the sharded run from the per-shard kernel times -- a projection, not a measurement"""
    W, nv = res["world"], res["Nv"]
    full, sh = res["full"], res["shards"]
    em = lambda k: k["estep"] + k["mstep"] + k["scale"]
    one = full["backproject"] + full["regularize"] + full["forward"] + em(full) + DEVICE_EM_MS   # (one GPU runs the device EM too)
    psf = max(k["backproject"] + k["forward"] + em(k) for k in sh)
    reg_full = max(k["regularize"] for k in sh)
    # replicated: all-reduce of addon|cmap (2 Nv floats: reduce-scatter + all-gather of the whole message), whole-volume update
    ar = 2 * collective_ms(2 * nv * 4, W)
    kind = "pvr" if str(res.get("workload", "")).startswith("PVR") else "svr"
    nex = HOST_EXCHANGES[kind]
    small = nex * HOST_EXCHANGE_MS + SMALL_COLLECTIVES[kind] * SMALL_COLLECTIVE_MS + DEVICE_EM_MS
    replicated = psf + ar + reg_full + small
    # slab: reduce-scatter of addon|cmap over the mask's voxels (+ halo planes), update of the rank's slab, all-gather of the volume
    mfrac = res.get("mask_fraction", 1.0)
    rs_ms = collective_ms(2 * nv * 4 * mfrac, W)
    ag_ms = collective_ms(nv * 4 * min(1.0, mfrac * 1.15), W)
    slab = psf + rs_ms + reg_full / W + ag_ms + small
    return dict(label="PROJECTION from one-GPU per-shard kernel times; no collective was run",
                assumptions=dict(xgmi_link_GBs_per_direction=XGMI_LINK_GBS, links_used=min(W - 1, 7), efficiency=XGMI_EFF,
                                 host_exchange_ms=HOST_EXCHANGE_MS, host_exchanges_per_step=nex, small_device_collectives_per_step=SMALL_COLLECTIVES[kind],
                                 small_device_collective_ms=SMALL_COLLECTIVE_MS, device_em_ms=DEVICE_EM_MS),
                one_gpu_kernels_ms=one, max_rank_psf_em_ms=psf, sum_rank_psf_ms=sum(k["backproject"] + k["forward"] for k in sh),
                shard_overhead=sum(k["backproject"] + k["forward"] for k in sh) / (full["backproject"] + full["forward"]),
                at_link_rates={name: project_at(res, (min(W - 1, 7) * XGMI_LINK_GBS * XGMI_EFF) if name.startswith("7 links") else rate)
                               for name, rate in LINK_RATES_GBS.items()},
                replicated=dict(allreduce_ms=ar, update_ms=reg_full, step_ms=replicated, speedup=one / replicated),
                slab=dict(reduce_scatter_ms=rs_ms, update_ms=reg_full / W, allgather_ms=ag_ms, step_ms=slab, speedup=one / slab))


def run(workload, world, reps=6, opts=(), only=None, layout=None):
    res = probe(workload, world, reps, opts, only, layout)
    if only is None or len(res["shards"]) == world:
        res["projection"] = project(res)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("world", type=int)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--out")
    ap.add_argument("--layout", choices=["spatial", "contiguous"], help="how the units are dealt to the ranks (sharding.shard_units); default: the product's")
    ap.add_argument("opts", nargs="*", help="engine options name=value")
    a = ap.parse_args()
    opts = [(o.split("=")[0], int(o.split("=")[1])) for o in a.opts]
    t0 = time.time()
    res = run(a.workload, a.world, a.reps, opts, layout=a.layout)
    res["wall_s"] = round(time.time() - t0, 1)
    line = json.dumps(res)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(line + "\n")
    print(line, flush=True)


if __name__ == "__main__":
    main()
