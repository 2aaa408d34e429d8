"""What a launcher of a sharded run needs on the host side (bench.py, the tests): contiguous unit ranges balanced by estimated
PSF work (csrc/svr_cli.cpp and csrc/pvr_cli.cpp do the same in C++), and a torch.distributed communicator for `--comm torch`
(the default is the C library's own RCCL binding, host.RcclComm)."""
from __future__ import annotations

import numpy as np


def shard_slices(active_per_slice, world):
    """Balanced contiguous slice ranges by active-pixel count; every slice is kept
    (the reference drops the last one and the remainder, RC.cu:1415,1440)."""
    a = np.asarray(active_per_slice, dtype=np.float64)
    ns = len(a)
    cum = np.concatenate([[0.0], np.cumsum(a + 1e-9)])
    bounds = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        b = int(np.searchsorted(cum, target, side="left"))
        b = min(max(b, bounds[-1]), ns)
        bounds.append(b)
    bounds.append(ns)
    return [(bounds[i], bounds[i + 1]) for i in range(world)]


DEFAULT_LAYOUT = "spatial"          # (SVR_SHARD_LAYOUT=contiguous brings the ranges of rounds 1-4 back: A/B measurements)


def shard_units(work, stack_index, world, layout=None):
    """-> (order, ranges): the numbering of a sharded run and every rank's contiguous range of it.

    order[k] = index, in the reference's order (stack after stack, slice after slice), of unit k of the sharded numbering;
    ranges[r] = (lo, hi) of that numbering.  The launcher uploads the units in the sharded numbering, tells the host objects `order`
    (svrh_set_unit_order / pvrh_set_unit_order: whatever the host arithmetic does in the reference's order it keeps doing in that
    order) and maps per-unit results back.

    layout "contiguous": order = identity, ranges balanced by work (rounds 1-4; what the reference's own sharding looks like,
    RC.cu:1413-1457).  layout "spatial" (round 5): rank r takes the r-th of `world` work-balanced segments of EVERY stack, so that
    a rank's units are neighbours in space: it then touches about 1 / world of the volume's (cell, plane) items per stack
    orientation instead of all the items of the stacks it holds -- less to stage and to combine per rank (csrc/svr_cell.inc), and
    every rank holds the same mix of orientations.  The cut points of a stack are placed against the running total over the stacks
    before it (error diffusion), so the ranks' totals differ by at most about one unit's work, whatever the number of stacks."""
    import os
    layout = layout or os.environ.get("SVR_SHARD_LAYOUT", DEFAULT_LAYOUT)
    w = np.asarray(work, np.float64) + 1e-9
    si = np.asarray(stack_index)
    n = len(w)
    if layout == "contiguous" or world <= 1:
        return np.arange(n, dtype=np.int32), shard_slices(w, world)
    if layout != "spatial":
        raise ValueError(layout)
    assert np.all(np.diff(si) >= 0), "units are expected stack after stack"
    parts = [[] for _ in range(world)]
    done = np.zeros(world + 1)                      # work handed out so far to the left of cut r, over the stacks so far
    total = 0.0
    for s in np.unique(si):
        idx = np.nonzero(si == s)[0]
        ws = w[idx]
        cum = np.concatenate([[0.0], np.cumsum(ws)])
        total += cum[-1]
        cuts = [0]
        for r in range(1, world):
            # cut r of this stack: the running total to the left of cut r comes closest to r / world of the running total
            want = total * r / world - done[r]
            b = int(np.argmin(np.abs(cum - want)))
            b = min(max(b, cuts[-1]), len(ws))
            cuts.append(b)
        cuts.append(len(ws))
        for r in range(world):
            parts[r].extend(idx[cuts[r]:cuts[r + 1]].tolist())
            done[r + 1] += cum[cuts[r + 1]]
        done[0] = 0.0
    order = np.array([i for q in parts for i in q], dtype=np.int32)
    ranges, at = [], 0
    for q in parts:
        ranges.append((at, at + len(q)))
        at += len(q)
    assert at == n and len(set(order.tolist())) == n
    return order, ranges


def patch_cost_weights(data_pixels_per_patch, patch_i2w, patch_t, recon_w2i):
    """Work of every patch for the sharding of the patch-based path: pixels that carry data x (1 + 0.2 n_x^2), n = the patch normal in
    volume axes (the same last factor as slice_cost_weights; csrc/pvr_cli.cpp does the same)."""
    act = np.asarray(data_pixels_per_patch, np.float64)
    i2w = np.asarray(patch_i2w, np.float64).reshape(-1, 4, 4)
    t = np.asarray(patch_t, np.float64).reshape(-1, 4, 4)
    w2i = np.asarray(recon_w2i, np.float64).reshape(4, 4)
    nrm = np.einsum("ij,sjk,sk->si", w2i[:3, :3], t[:, :3, :3], i2w[:, :3, 2])
    n2 = np.maximum((nrm ** 2).sum(1), 1e-24)
    return act * (1.0 + 0.2 * nrm[:, 0] ** 2 / n2)


def slice_cost_weights(active_per_slice, slice_i2w, slice_t, recon_w2i, slice_dim, voxel):
    """Estimated PSF work of every slice for the sharding: active pixels x (9.4 + live planes) x (1 + 0.2 n_x^2).  A (pixel, plane) unit of
    the owned axis e (the volume axis y or z closest to the slice normal n) is evaluated in full unless all of its taps
    lie further than 5.1 sigma_z from the slice plane: |d n_e| - 8 (|n_x| + |n_o|) > 5.1 sigma_z (csrc/svr_hip.hip,
    unit_is_dead).  Axial / coronal slices keep ~12 of their 16 planes, sagittal ones (normal along x, the axis of the
    sequential epsilon-chain, which cannot be owned) all 16.  Measured per stack on P4 (scatter + gather, ns per active
    pixel): axial 8.1, coronal 8.2, sagittal 9.6, in-plane rotated axial 8.8 -- the constant 9.4 is fitted to the first three.  Round 4 measured what a
    RANK pays (tools/shard_probe.py: 8 ranks, each with its own range, one after the other on one GPU): sagittal / axial per pixel 1.26
    on S8 -- where the thick slices keep 15.4 of 16 planes alive whatever their orientation, so the live planes explain nothing -- and
    1.37 on P4, and the ranks holding the sagittal stacks were the slowest by 17 %.  A run of a slice whose normal is the x axis
    spans a band of 8 centre planes (svr_cell.inc): its pixels are live on 16 of the 23 planes the run visits.  Hence the last
    factor (S8 1.02 x 1.2, P4 1.19 x 1.2); sharding by pixel count alone leaves the ranks that hold the sagittal stack with 1.19x the work."""
    act = np.asarray(active_per_slice, np.float64)
    i2w = np.asarray(slice_i2w, np.float64).reshape(-1, 4, 4)
    t = np.asarray(slice_t, np.float64).reshape(-1, 4, 4)
    w2i = np.asarray(recon_w2i, np.float64).reshape(4, 4)
    nrm = np.einsum("ij,sjk,sk->si", w2i[:3, :3], t[:, :3, :3], i2w[:, :3, 2])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-12)
    ax, ay, az = np.abs(nrm[:, 0]), np.abs(nrm[:, 1]), np.abs(nrm[:, 2])
    ne, no = np.maximum(ay, az), np.minimum(ay, az)
    sigma = np.asarray(slice_dim, np.float64).reshape(-1, 3)[:, 2] / 2.3548 / float(voxel)
    live = np.minimum(16.0, 2.0 * (5.1 * sigma + 8.0 * (ax + no)) / np.maximum(ne, 1e-3) + 1.0)
    return act * (9.4 + live) * (1.0 + 0.2 * ax * ax)


class TorchComm:
    """torch.distributed communicator: RCCL ("nccl") on GPUs, gloo in the CPU tests.

    Volume all-reduce runs on the device buffer in place (one float[2*Nv] message per pass over
    xGMI); small host vectors go through the same backend.
    """

    def __init__(self, device=None, slabs=False):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.slabs = slabs            # the volume update by z-slabs on the CPU stand-in engines (slab_update_numpy)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        self._views = {}

    def _host(self, a, op):
        t = self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=op)
        return t.cpu().numpy()

    def allreduce_sum(self, a):
        return self._host(a, self.dist.ReduceOp.SUM)

    def allreduce_min(self, a):
        return self._host(a, self.dist.ReduceOp.MIN)

    def allreduce_max(self, a):
        return self._host(a, self.dist.ReduceOp.MAX)

    def allgather_slices(self, local, counts):
        n = max(counts)
        t = self.torch.zeros(n, dtype=self.torch.float32)
        t[: len(local)] = self.torch.from_numpy(np.asarray(local, np.float32))
        if self.device is not None:
            t = t.to(self.device)
        outs = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(outs, counts)])

    def allreduce_volume_pair(self, engine, which):
        """In-place all-reduce of {recon|volw} (which=0) or {addon|cmap} (which=2)."""
        if hasattr(engine, "volume_pair_tensor"):      # CPU stand-in engines used by the gloo tests
            t = engine.volume_pair_tensor(which)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            engine.volume_pair_commit(which, t)
            return
        key = (id(engine), which)
        if key not in self._views:
            self._views[key] = _device_view(self.torch, engine.device_ptr(which), 2 * int(np.prod(engine.vsize)),
                                            self.device)
        self.dist.all_reduce(self._views[key], op=self.dist.ReduceOp.SUM)
        self.torch.cuda.synchronize()


class _CAI:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def _device_view(torch, ptr, n, device):
    """Zero-copy torch view of an engine-owned device buffer."""
    return torch.as_tensor(_CAI(ptr, n), device=device)


