"""Host algorithm object: the `*GPU()` operator surface of the reference's irtkReconstruction
(source/reconstructionGPU2/irtkReconstructionGPU.cc, "RG.cc") driving one engine per process.

Method names, state names and arithmetic follow RG.cc so a reference user finds the same
operators: InitializeEMGPU (RG.cc:2921-2953), InitializeEMValuesGPU (2905-2919),
GaussianReconstructionGPU (2695-2762), SimulateSlicesGPU (1163-1175),
InitializeRobustStatisticsGPU (2988-3019), EStepGPU (3184-3440), ScaleGPU (3751-3757),
SuperresolutionGPU (4024-4036), MStepGPU (4214-4223), MaskVolumeGPU (5319-5323), and the
reconstruction loop of reconstruction.cc:816-1140 in `reconstruct()`.

Multi-GPU (no reference equivalent; replaces GPUWorker.cpp): slices are sharded over ranks, every
rank keeps the whole volume, the volume accumulators are all-reduced once per scatter pass, the
5 M-step scalars once per M-step, and the per-slice vectors (potentials, scales) are all-gathered
so the slice-level EM (host code) runs identically on every rank.
"""
from __future__ import annotations

import math

import numpy as np

from fetalreconstruction_amd.sharding import TorchComm, patch_cost_weights, shard_slices, slice_cost_weights  # noqa: F401  (re-exported for the tests)


def _seqsum(a):
    """Sequential (left-to-right) double sum, like a `for` loop with `sum += x`."""
    return float(np.cumsum(a, dtype=np.float64)[-1]) if len(a) else 0.0


class LocalComm:
    """Single-process communicator (world_size 1)."""

    rank, world = 0, 1

    def allreduce_volume_pair(self, engine, which):
        pass

    def allreduce_sum(self, a):
        return a

    def allreduce_min(self, a):
        return a

    def allreduce_max(self, a):
        return a

    def allgather_slices(self, local, counts):
        return local


def slab_plan_numpy(mask_zyx, world):
    """The plan of csrc/svr_slab.inc (slab_lists + slab_plan) in numpy: index lists of the mask's voxels and of its 3 x 3 x 3
    dilation in natural order, slabs of equal mask-voxel count, per rank the index range of its slab + one halo plane either
    side (reduce-scatter) and of its slab in the dilated list (all-gather)."""
    m = np.asarray(mask_zyx) != 0
    vz = m.shape[0]
    d = np.zeros_like(m)
    p = np.pad(m, 1)
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                d |= p[dz:dz + vz, dy:dy + m.shape[1], dx:dx + m.shape[2]]
    midx, didx = np.flatnonzero(m), np.flatnonzero(d)
    mcum = np.concatenate([[0], np.cumsum(m.reshape(vz, -1).sum(1))]).astype(np.int64)
    dcum = np.concatenate([[0], np.cumsum(d.reshape(vz, -1).sum(1))]).astype(np.int64)
    nm = int(mcum[-1])
    zb = [0]
    for r in range(1, world):
        target = nm * r // world
        z = zb[-1]
        while z < vz and mcum[z] < target:
            z += 1
        zb.append(z)
    zb.append(vz)
    rs, ag = [], []
    for r in range(world):
        lo, hi = max(0, zb[r] - 1), min(vz, zb[r + 1] + 1)
        rs.append((int(mcum[lo]), 0 if zb[r] >= zb[r + 1] else int(mcum[hi] - mcum[lo])))
        ag.append((int(dcum[zb[r]]), int(dcum[zb[r + 1]] - dcum[zb[r]])))
    return dict(midx=midx, didx=didx, zb=zb, rs=rs, ag=ag, rs_chunk=max(1, max(c for _, c in rs)), ag_chunk=max(1, max(c for _, c in ag)))


def slab_update_numpy(e, comm, args):
    """reduce-scatter of addon | cmap at the mask's voxels -> the rank's planes of the volume update -> all-gather of the new volume,
    on an engine that keeps its volumes as numpy arrays (the oracle): what Shard::update does with the svr_slab_* entry points.  The
    reduce-scatter is an all-reduce of the send buffer of which a rank keeps its chunk (gloo has no reduce-scatter): the sums and
    their order are those of the replicated path's all-reduce."""
    torch, dist = comm.torch, comm.dist
    W, R = comm.world, comm.rank
    vx, vy, vz = (int(v) for v in e.vsize)
    plan = getattr(e, "_slab_plan", None)
    if plan is None or plan["world"] != W:
        plan = slab_plan_numpy(np.asarray(e.mask).reshape(vz, vy, vx), W)
        plan["world"] = W
        e._slab_plan = plan
    CH, DCH = plan["rs_chunk"], plan["ag_chunk"]
    send = np.zeros((W, 2, CH), np.float32)
    for r, (st, cnt) in enumerate(plan["rs"]):
        idx = plan["midx"][st:st + cnt]
        send[r, 0, :cnt] = e.addon[idx]
        send[r, 1, :cnt] = e.cmap[idx]
    t = torch.from_numpy(send)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    st, cnt = plan["rs"][R]
    idx = plan["midx"][st:st + cnt]
    e.addon[idx] = send[R, 0, :cnt]
    e.cmap[idx] = send[R, 1, :cnt]
    e.SuperresolutionUpdate(*args)                 # whole volume; only this rank's planes are kept (a plane depends on its neighbours only)
    new = e.recon.copy()
    plane = vx * vy
    z0, z1 = plan["zb"][R], plan["zb"][R + 1]
    mine = np.zeros(DCH, np.float32)
    st, cnt = plan["ag"][R]
    mine[:cnt] = new[plan["didx"][st:st + cnt]]
    outs = [torch.zeros(DCH, dtype=torch.float32) for _ in range(W)]
    dist.all_gather(outs, torch.from_numpy(mine))
    e.recon[...] = 0
    e.recon[z0 * plane:z1 * plane] = new[z0 * plane:z1 * plane]
    for r, (st, cnt) in enumerate(plan["ag"]):
        if r != R:
            e.recon[plan["didx"][st:st + cnt]] = outs[r].numpy()[:cnt]


class irtkReconstruction:
    """GPU-path operator surface of irtkReconstruction for one rank's shard of the slices."""

    def __init__(self, engine, n_slices_global, slice_range=None, comm=None, max_intensity=1.0,
                 min_intensity=0.0, debug=False):
        self.reconstructionGPU = engine
        self.comm = comm or LocalComm()
        self.ns = int(n_slices_global)
        self.lo, self.hi = slice_range if slice_range is not None else (0, self.ns)
        self.counts = None
        # RG.cc:159-221
        self._step = 0.0001
        self._debug = debug
        self._quality_factor = 2
        self._sigma_bias = 12
        self._sigma_s_gpu = 0.025
        self._mix_s_gpu = 0.9
        self._mix_gpu = 0.9
        self._delta = 1.0
        self._lambda = 0.1
        self._alpha = (0.05 / self._lambda) * self._delta * self._delta
        self._low_intensity_cutoff = 0.01
        self._global_bias_correction = False
        self._adaptive = False
        self._max_intensity = float(max_intensity)
        self._min_intensity = float(min_intensity)
        self._force_excluded = []
        self._small_slices = []
        self._disableBiasC = True      # reconstruction.cc:121,202: the CLI can never switch it on
        self._intensity_matching = True  # reconstruction.cc:114,183: --no_intensity_matching 0 switches Bias / Scale / NormaliseBias off
        self._scale_gpu = np.ones(self.ns, np.float32)
        self._slice_weight_gpu = np.ones(self.ns, np.float32)
        self._slice_inside_gpu = np.ones(self.ns, bool)
        self._sigma_gpu = 0.0
        self._m_gpu = 0.0
        self._mean_s_gpu = 0.0
        self._mean_s2_gpu = 0.0
        self._sigma_s2_gpu = 0.025

    # -- helpers ---------------------------------------------------------------------------
    def _local(self, v):
        return np.ascontiguousarray(v[self.lo:self.hi], np.float32)

    def _gather(self, local):
        if self.comm.world == 1:
            return np.asarray(local, np.float32)
        if self.counts is None:
            c = self.comm.allreduce_sum(np.eye(self.comm.world)[self.comm.rank] * (self.hi - self.lo))
            self.counts = [int(round(x)) for x in c]
        return self.comm.allgather_slices(local, self.counts).astype(np.float32)

    def SetSmoothingParameters(self, delta, lam):
        """RG.h:605-612"""
        self._delta = delta
        self._lambda = lam * delta * delta
        self._alpha = 0.05 / lam
        if self._alpha > 1:
            self._alpha = 1

    def SpeedupOn(self):
        self._quality_factor = 1

    def SpeedupOff(self):
        self._quality_factor = 2

    def SetForceExcludedSlices(self, force_excluded):
        self._force_excluded = list(force_excluded)

    # -- operators -------------------------------------------------------------------------
    def InitializeEMValuesGPU(self):
        """RG.cc:2905-2919"""
        self._slice_weight_gpu = np.ones(self.ns, np.float32)
        self._scale_gpu = np.ones(self.ns, np.float32)
        self.reconstructionGPU.UpdateScaleVector(self._local(self._scale_gpu), self._local(self._slice_weight_gpu))
        self.reconstructionGPU.InitializeEMValues()

    def InitializeEMGPU(self):
        """RG.cc:2921-2953 (the intensity range is found by the caller on the host slices)"""
        self.InitializeEMValuesGPU()

    def GaussianReconstructionGPU(self):
        """RG.cc:2695-2762.  voxel_num has one entry per device in the reference and its median
        indexes out of range for one device (RG.cc:2714-2726), so no slice is
        ever classified as small on the GPU path: _small_slices stays empty."""
        e = self.reconstructionGPU
        if self.comm.world == 1:
            e.GaussianReconstruction()
        else:
            e.GaussianReconstructionLocal()
            self.comm.allreduce_volume_pair(e, 0)
            e.GaussianReconstructionFinish()
        self._small_slices = []

    def SimulateSlicesGPU(self):
        """RG.cc:1163-1175"""
        inside = self.reconstructionGPU.SimulateSlices()
        self._slice_inside_gpu = self._gather(np.asarray(inside, np.float32)) > 0.5

    def InitializeRobustStatisticsGPU(self):
        """RG.cc:2988-3019"""
        e = self.reconstructionGPU
        if self.comm.world == 1:
            self._sigma_gpu = e.InitializeRobustStatistics()
        else:
            s = self.comm.allreduce_sum(e.RobustStatisticsSums())
            self._sigma_gpu = float(np.float32(s[0]) / np.float32(s[1]))
        self._slice_weight_gpu[~self._slice_inside_gpu] = 0
        for i in self._force_excluded:
            self._slice_weight_gpu[i] = 0
        self._sigma_s_gpu = 0.025
        self._mix_gpu = 0.9
        self._mix_s_gpu = 0.9
        # (float)(1.0f / (2.1f * _max_intensity - 1.9f * _min_intensity)): float literals, double members
        self._m_gpu = float(np.float32(1.0 / (float(np.float32(2.1)) * self._max_intensity -
                                               float(np.float32(1.9)) * self._min_intensity)))
        e.UpdateScaleVector(self._local(self._scale_gpu), self._local(self._slice_weight_gpu))

    def _G(self, x, s):
        """RG.h:529-532"""
        return self._step * math.exp(-x * x / (2 * s)) / (math.sqrt(6.28 * s))

    def EStepGPU(self):
        """RG.cc:3184-3440: voxel posteriors on the GPU, slice-level EM on the host.

        Vectorised with numpy; every reduction is a sequential double sum (`_seqsum`) like the
        reference's `sum += ...` loops, so the result is the loop's bit for bit."""
        f32 = np.float32
        pot = self._gather(self.reconstructionGPU.EStep(self._m_gpu, self._sigma_gpu, self._mix_gpu))
        pot = pot.astype(np.float32).copy()
        w = self._slice_weight_gpu
        for i in self._force_excluded:
            pot[i] = -1
        for i in self._small_slices:
            pot[i] = -1
        pot[(self._scale_gpu < 0.2) | (self._scale_gpu > 5)] = -1           # RG.cc:3212-3215
        ok = pot >= 0
        p64, w64 = pot[ok].astype(np.float64), w[ok].astype(np.float64)
        s, den = _seqsum(p64 * w64), _seqsum(w64)
        s2, den2 = _seqsum(p64 * (1.0 - w64)), _seqsum(1.0 - w64)
        maxs = max(0.0, float(p64.max())) if p64.size else 0.0
        mins = min(1.0, float(p64.min())) if p64.size else 1.0
        mean_s = float(f32(s / den)) if den > 0 else float(f32(mins))
        mean_s2 = float(f32(s2 / den2)) if den2 > 0 else float(f32((maxs + mean_s) / 2.0))
        s = _seqsum((p64 - mean_s) * (p64 - mean_s) * w64)
        s2 = _seqsum((p64 - mean_s2) * (p64 - mean_s2) * (1 - w64))
        floor = self._step * self._step / 6.28
        if s > 0 and den > 0:
            sigma_s = float(f32(s / den))
            if sigma_s < floor:
                sigma_s = float(f32(floor))
        else:
            sigma_s = float(f32(0.025))
        if s2 > 0 and den2 > 0:
            sigma_s2 = float(f32(s2 / den2))
            if sigma_s2 < floor:
                sigma_s2 = float(f32(floor))
        else:
            sigma_s2 = float(f32(f32(f32(mean_s2) - f32(mean_s)) * f32(f32(mean_s2) - f32(mean_s)) / f32(4)))
            if sigma_s2 < floor:
                sigma_s2 = float(f32(floor))
        mix_s = float(f32(self._mix_s_gpu))
        # slice weights, RG.cc:3365-3404
        neg = pot == -1
        if den <= 0 or mean_s2 <= mean_s:
            neww = np.ones(self.ns, np.float32)
        else:
            p = pot.astype(np.float64)
            with np.errstate(over="ignore", under="ignore", invalid="ignore", divide="ignore"):
                g1 = np.where(p < mean_s2, self._step * np.exp(-(p - mean_s) ** 2 / (2 * sigma_s)) / math.sqrt(6.28 * sigma_s), 0.0)
                g2 = np.where(p > mean_s, self._step * np.exp(-(p - mean_s2) ** 2 / (2 * sigma_s2)) / math.sqrt(6.28 * sigma_s2), 0.0)
                like = g1 * mix_s + g2 * (1 - mix_s)
                neww = np.where(like > 0, g1 * mix_s / np.where(like > 0, like, 1.0), w.astype(np.float64))
            zero_like = ~(like > 0)
            neww = np.where(zero_like & (p <= mean_s), 1.0, neww)
            neww = np.where(zero_like & (p >= mean_s2), 0.0, neww)
            neww = np.where(zero_like & (p < mean_s2) & (p > mean_s), 1.0, neww)
            neww = neww.astype(np.float32)
        neww[neg] = 0
        w[...] = neww
        num = int(ok.sum())
        self._mix_s_gpu = float(f32(_seqsum(w[ok].astype(np.float64)) / num)) if num > 0 else 0.9
        self._mean_s_gpu, self._mean_s2_gpu = mean_s, mean_s2
        self._sigma_s_gpu, self._sigma_s2_gpu = sigma_s, sigma_s2
        self._slice_potential_gpu = pot
        self.reconstructionGPU.UpdateSliceWeights(self._local(w))

    def ScaleGPU(self):
        """RG.cc:3751-3757"""
        self._scale_gpu = self._gather(self.reconstructionGPU.CalculateScaleVector())

    def SuperresolutionGPU(self, it):
        """RG.cc:4024-4036"""
        e = self.reconstructionGPU
        if self.comm.world == 1:
            e.Superresolution(it, self._local(self._slice_weight_gpu), self._adaptive, self._alpha,
                              self._min_intensity, self._max_intensity, self._delta, self._lambda,
                              self._global_bias_correction, self._sigma_bias, self._low_intensity_cutoff)
        else:
            e.SuperresolutionBackproject(self._local(self._slice_weight_gpu))
            args = (self._adaptive, self._alpha, self._min_intensity, self._max_intensity, self._delta, self._lambda)
            if getattr(self.comm, "slabs", False) and hasattr(e, "addon_cmap"):
                slab_update_numpy(e, self.comm, args)            # csrc/svr_slab.inc restated on the CPU stand-in engines (gloo tests)
            else:
                self.comm.allreduce_volume_pair(e, 2)
                e.SuperresolutionUpdate(*args)

    def MStepGPU(self, it):
        """RG.cc:4214-4223 + Reconstruction::MStep host part (RC.cu:3016-3071)"""
        e = self.reconstructionGPU
        if self.comm.world == 1:
            self._sigma_gpu, self._mix_gpu, self._m_gpu = e.MStep(it, self._step, self._sigma_gpu, self._mix_gpu)
            return
        f32 = np.float32
        s5 = e.MStepSums()
        tot = self.comm.allreduce_sum(s5[:3])
        mn = self.comm.allreduce_min(s5[3:4])[0]
        mx = self.comm.allreduce_max(s5[4:5])[0]
        sigma, mix, num = f32(tot[0]), f32(tot[1]), f32(tot[2])
        fmax, fmin = np.finfo(np.float32).max, np.finfo(np.float32).tiny
        min_ = min(fmax, f32(mn))
        max_ = max(fmin, f32(mx))
        step = f32(self._step)
        if mix > 0:
            self._sigma_gpu = float(sigma / mix)
        if f32(self._sigma_gpu) < step * step / f32(6.28):
            self._sigma_gpu = float(step * step / f32(6.28))
        if it > 1:
            self._mix_gpu = float(mix / num)
        self._m_gpu = float(f32(1.0) / (f32(max_) - f32(min_)))

    def BiasGPU(self):
        """RG.cc:3904-3913"""
        self.reconstructionGPU.CorrectBias(self._sigma_bias, self._global_bias_correction)

    def NormaliseBiasGPU(self, it):
        """RG.cc:4653-4655 (single rank; the sharded split is svr_normalise_bias_local/_finish)"""
        self.reconstructionGPU.NormaliseBias(it, self._sigma_bias)

    def MaskVolumeGPU(self):
        self.reconstructionGPU.maskVolume()

    def ScaleVolumeGPU(self):
        e = self.reconstructionGPU
        if self.comm.world == 1:
            e.ScaleVolume()
        else:
            s = self.comm.allreduce_sum(e.ScaleVolumeSums())
            e.ScaleVolumeApply(float(np.float32(s[0] / s[1])))

    # -- the reconstruction part of main()'s loop (reconstruction.cc:895-1140) ----------------
    def reconstruct_iteration(self, rec_iterations, on_sr_iteration=None):
        """One outer iteration after registration: Gaussian init, robust-statistics init and
        `rec_iterations` super-resolution iterations.  `on_sr_iteration(i)` is a timing hook."""
        self.InitializeEMValuesGPU()
        self.GaussianReconstructionGPU()
        self.SimulateSlicesGPU()
        self.InitializeRobustStatisticsGPU()
        self.EStepGPU()
        for i in range(rec_iterations):
            self.sr_iteration(i)
            if on_sr_iteration:
                on_sr_iteration(i)
        self.MaskVolumeGPU()

    def sr_iteration(self, i):
        """The hot loop body, reconstruction.cc:1013-1108 with bias correction off."""
        if self._intensity_matching:                              # reconstruction.cc:1018-1045
            if not self._disableBiasC and self._sigma_bias > 0:  # reconstruction.cc:1032-1037
                self.BiasGPU()
            self.ScaleGPU()
        self.SuperresolutionGPU(i + 1)
        if self._intensity_matching and not self._disableBiasC and self._sigma_bias > 0 and not self._global_bias_correction:
            self.NormaliseBiasGPU(i)                              # reconstruction.cc:1066-1076
        self.SimulateSlicesGPU()
        self.MStepGPU(i + 1)
        self.EStepGPU()
