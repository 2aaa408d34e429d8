"""Host side of the patch-to-volume reconstruction (PVR): patch extraction and the
reconstruction loop of irtkPatchBasedReconstruction, driving one engine context with option pvr=1.

Mirrors (R2 = /root/reference/source/reconstructionGPU2):
  * PatchBasedObject<T>::generate2DPatches           R2/include/patchBasedObject.cuh:176-342
  * irtkPatchBasedReconstruction<T>::run (the loop)  R2/irtkPatchBasedReconstruction.cpp:426-593
  * patchBasedRobustStatistics_gpu<T>::{initializeEMValues, InitializeRobustStatistics, EStep, MStep,
    Scale}                                            R2/patchBasedRobustStatistics_gpu.cu:78-95, 224-556,
                                                      570-640, 672-745, 793-845
  * patchBasedSuperresolution_gpu<T>::{run, regularize} with its fixed delta = 1, lambda = 0.1,
    alpha = 0.05 / lambda * delta^2                   R2/patchBasedSuperresolution_gpu.cu:113-336
Patches are handed to the engine as the slices of its padded grid `[nPatches][pY][pX]`
(R2/include/patchBasedVolume.cuh:108-194); per-patch scale and patchWeight are the engine's scale /
slice-weight vectors.  Patch-to-volume registration (runHybrid) is not part of this module.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass

import numpy as np

from fetalreconstruction_amd import geometry as geo
from fetalreconstruction_amd.phantom import Problem, Stack  # noqa: F401




def _snap(v):
    """A patch pixel sits exactly on a slice pixel (and, for a mask drawn on the stack's grid, on a mask voxel): its
    coordinate is an integer in exact arithmetic, and the reference truncates what the double arithmetic makes of it
    (patchBasedObject.cuh:262-277), so on an oblique grid 19.999999999999996 reads pixel 19 or 20 depending on the last
    bit.  Here a coordinate within 1e-6 of an integer IS that integer; everything else is untouched."""
    r = np.rint(v)
    return np.where(np.abs(v - r) < 1e-6, r, v)


def generate2DPatches(stack: Stack, mask: np.ndarray, mask_attr: geo.ImageAttributes, pbbsize, stride, with_origins=False):
    """patchBasedObject.cuh:176-342.  Returns (patches float32 [n][pY][pX], I2W [n][16], W2I [n][16],
    total_pixels).  A patch starts as an all-zero image (irtkGenericImage(sattr)); pixels whose position
    is inside the slice and inside the mask (mask > 0, no stack transformation applied) are copied; the
    patch is kept when more than a third of its pixels are set to something other than 0 / -1."""
    a = stack.attr
    px, py = int(pbbsize[0]), int(pbbsize[1])
    sx, sy = int(stride[0]), int(stride[1])
    s_i2w = geo.image_to_world(a)
    m_w2i = geo.world_to_image(mask_attr)
    mz, my, mx = mask.shape
    jj, ii = np.meshgrid(np.arange(py), np.arange(px), indexing="ij")
    pix = np.stack([ii, jj, np.zeros_like(ii), np.ones_like(ii)], -1).astype(np.float64)   # [pY][pX][4]
    out, i2ws, w2is, origins = [], [], [], []
    total = 0
    for z in range(a.nz):
        sl_attr = geo.ImageAttributes(a.nx, a.ny, 1, a.dx, a.dy, stack.thickness * 2, a.xaxis, a.yaxis, a.zaxis)
        sl_attr.origin = geo.region_origin(a, 0, 0, z, sl_attr)                # GetRegion + PutPixelSize :204-205
        sl_i2w, sl_w2i = geo.image_to_world(sl_attr), geo.world_to_image(sl_attr)
        p0 = geo.ImageAttributes(px, py, 1, a.dx, a.dy, stack.thickness * 2, a.xaxis, a.yaxis, a.zaxis)
        p0_first = geo.apply_points(geo.image_to_world(p0), np.array([0.0, 0.0, 0.0, 1.0]))
        for y in range(0, a.ny + py, sy):
            for x in range(0, a.nx + px, sx):
                first = geo.apply_points(sl_i2w, np.array([float(x), float(y), 0.0, 1.0]))
                pa = copy.copy(p0)
                pa.origin = (first - p0_first)[:3]                              # :232-246
                p_i2w = geo.image_to_world(pa)
                w = geo.apply_points(p_i2w, pix)                                # patch.ImageToWorld, then slice / mask WorldToImage
                q = geo.apply_points(sl_w2i, w)
                xx, yy = _snap(q[..., 0]), _snap(q[..., 1])
                qm = geo.apply_points(m_w2i, w)
                x1, y1, z1 = _snap(qm[..., 0]), _snap(qm[..., 1]), _snap(qm[..., 2])
                ok = (xx >= 0) & (yy >= 0) & (xx < a.nx) & (yy < a.ny)
                ok &= (x1 >= 0) & (y1 >= 0) & (z1 >= 0) & (x1 < mx) & (y1 < my) & (z1 < mz)
                xi, yi = np.clip(xx.astype(int), 0, a.nx - 1), np.clip(yy.astype(int), 0, a.ny - 1)   # int truncation
                mi = mask[np.clip(z1.astype(int), 0, mz - 1), np.clip(y1.astype(int), 0, my - 1),
                          np.clip(x1.astype(int), 0, mx - 1)]
                ok &= mi > 0
                patch = np.where(ok, stack.data[z][yi, xi], 0.0)
                set_count = int((ok & (patch != 0) & (patch != -1)).sum())
                if set_count > np.float32(1.0) / np.float32(3.0) * np.float32(py) * np.float32(px):    # :318
                    total += set_count
                    out.append(patch.astype(np.float32))
                    i2ws.append(geo.to_matrix4(p_i2w))
                    w2is.append(geo.to_matrix4(geo.world_to_image(pa)))
                    origins.append(np.asarray(pa.origin, np.float64).copy())
    n = len(out)
    res = (np.stack(out) if n else np.zeros((0, py, px), np.float32),
           np.stack(i2ws) if n else np.zeros((0, 16), np.float32),
           np.stack(w2is) if n else np.zeros((0, 16), np.float32), total)
    return res + (np.stack(origins) if n else np.zeros((0, 3)),) if with_origins else res


def make_pvr_problem(stacks, mask, mask_attr, recon_attr, recon_mask, pbbsize=(32, 32), stride=(16, 16), name="pvr", superpixel=False,
                     full_slices=False):
    """PatchBasedVolume<T>::init for every stack (irtkPatchBasedReconstruction.cpp:385-399) packed into
    one Problem: slices = patches, slice dims = the stack's voxel size `getDim()` (z = stack spacing,
    R2/patchBasedPSFReconstruction_gpu.cu:67), T = the stack transformation.  full_slices = --useFullSlices: the patch is the
    whole slice and the stride steps past it (patchBasedObject.cuh:183-189), one patch per slice that covers the mask by a third;
    stacks of different in-plane sizes share one grid padded with -1."""
    P, I, W, T, TI, D, SI, counts, RI, MO, MI, AT, SM = [], [], [], [], [], [], [], [], [], [], [], [], []
    gx = max(st.attr.nx for st in stacks) if full_slices else 0
    gy = max(st.attr.ny for st in stacks) if full_slices else 0
    for k, st in enumerate(stacks):
        if full_slices and not superpixel:
            pbb = (st.attr.nx, st.attr.ny)
            p0, i2w, w2i, _, org = generate2DPatches(st, mask, mask_attr, pbb, (pbb[0] + 1, pbb[1] + 1), with_origins=True)
            p = np.full((len(p0), gy, gx), -1.0, np.float32)
            p[:, :pbb[1], :pbb[0]] = p0
        elif superpixel:      # pbbsize = --spxSize, stride = --spxExtend (pvrmain:291-296); patches of 64x64 with a mask each
            from . import slic
            p, i2w, w2i, sm, org, _ = slic.generate2DSuperpixelPatches(st, mask, mask_attr, pbbsize, stride[0])
            SM.append(sm)
            pbb = (p.shape[2], p.shape[1])
        else:
            p, i2w, w2i, _, org = generate2DPatches(st, mask, mask_attr, pbbsize, stride, with_origins=True)
            pbb = pbbsize
        n = len(p)
        # the origin-reset matrices of a patch (patchBasedObject.cuh:285-304): Mo = translation by the patch origin,
        # RI2W = the image-to-world matrix of the same patch with its origin at 0
        a0 = geo.ImageAttributes(int(pbb[0]), int(pbb[1]), 1, st.attr.dx, st.attr.dy, st.thickness * 2, st.attr.xaxis, st.attr.yaxis,
                                 st.attr.zaxis)
        ri = geo.to_matrix4(geo.image_to_world(a0))
        for o in org:
            mo = np.eye(4)
            mo[:3, 3] = o
            RI.append(ri); MO.append(geo.to_matrix4(mo)); MI.append(geo.to_matrix4(np.linalg.inv(mo)))
            at = copy.copy(a0)
            at.origin = np.asarray(o, np.float64).copy()
            AT.append(at)
        counts.append(n)
        P.append(p); I.append(i2w); W.append(w2i)
        T.append(np.tile(geo.to_matrix4(st.transformation), (n, 1)))
        TI.append(np.tile(geo.to_matrix4(np.linalg.inv(st.transformation)), (n, 1)))
        D.append(np.tile(np.array([st.attr.dx, st.attr.dy, st.attr.dz], np.float32), (n, 1)))
        SI.append(np.full(n, k, np.int32))
    patches = np.concatenate(P)
    n = len(patches)
    pos = patches[patches > 0]
    rd = (recon_attr.dx, recon_attr.dy, recon_attr.dz)
    prob = Problem(
        vsize=(recon_attr.nx, recon_attr.ny, recon_attr.nz), vdim=rd,
        recon_i2w=geo.to_matrix4(geo.image_to_world(recon_attr)),
        recon_w2i=geo.to_matrix4(geo.world_to_image(recon_attr)),
        mask=np.ascontiguousarray(recon_mask, np.float32), slices=patches, slice_i2w=np.concatenate(I), slice_w2i=np.concatenate(W),
        slice_t=np.concatenate(T), slice_tinv=np.concatenate(TI), slice_dim=np.concatenate(D),
        sizes_x=np.full(n, patches.shape[2], np.int32), sizes_y=np.full(n, patches.shape[1], np.int32),
        stack_index=np.concatenate(SI), psf_c0=geo.psf_centre_offset(rd),
        min_intensity=float(pos.min()) if pos.size else 0.0, max_intensity=float(pos.max()) if pos.size else 1.0,
        name=name)
    prob.patches_per_stack = counts
    prob.patch_ri2w = np.stack(RI) if RI else np.zeros((0, 16), np.float32)
    prob.patch_mo = np.stack(MO) if MO else np.zeros((0, 16), np.float32)
    prob.patch_invmo = np.stack(MI) if MI else np.zeros((0, 16), np.float32)
    prob.slice_attr = AT                                   # the patches as images (the targets of the patch-to-volume registration)
    prob.spx_masks = np.concatenate(SM) if SM else None    # [n][4096] '1' / 0, 64 wide (ImagePatch2D.cuh:51)
    return prob


def _G(x, s, step=np.float32(0.00001)):
    """G_<float> of patchBasedRobustStatistics_gpu.cu:97-101 with __step = 0.00001f, evaluated in float."""
    x, s = np.float32(x), np.float32(s)
    return np.float32(step * np.exp(np.float32(-x * x / (np.float32(2.0) * s))) / np.sqrt(np.float32(6.28) * s))


class irtkPatchBasedReconstruction:
    """The reconstruction part of irtkPatchBasedReconstruction<T>::run
    (irtkPatchBasedReconstruction.cpp:445-593) on one engine (`engine.Reconstruction` with option pvr=1,
    or the oracle twin).  Members carry the reference's names (m_sigma_gpu ...)."""

    def __init__(self, engine, patches_per_stack, min_intensity, max_intensity, adaptive=False, patch_range=None, comm=None):
        """patch_range = (lo, hi), comm: the engine holds the patches [lo, hi) of the global numbering and the ranks exchange
        like csrc/pvr_host.cpp does (the gloo tests run this driver on oracle engines); patches_per_stack stays global."""
        self.e = engine
        self.counts = [int(c) for c in patches_per_stack]
        self.n = int(sum(self.counts))
        self.lo, self.hi = patch_range if patch_range is not None else (0, self.n)
        self.comm = comm if comm is not None and comm.world > 1 else None
        self._scale_stale = False
        self.m_min_intensity, self.m_max_intensity = float(min_intensity), float(max_intensity)
        self.m_adaptive = bool(adaptive)
        self.m_delta = np.float32(1.0)                                   # patchBasedSuperresolution_gpu.cu:291-295
        self.m_lambda = np.float32(0.1)
        self.m_alpha = np.float32(np.float32(0.05) / self.m_lambda) * self.m_delta * self.m_delta
        self.m_step = np.float32(0.0001)                                  # T m_step, patchBasedRobustStatistics_gpu.cu:877
        self.scale = np.ones(self.n, np.float32)
        self.patch_weight = np.ones(self.n, np.float32)
        self.m_sigma_gpu = self.m_mix_gpu = self.m_m_gpu = np.float32(0)
        self.m_sigma_s_gpu = self.m_mix_s_gpu = np.float32(0)
        self.m_mean_s_gpu = self.m_mean_s2_gpu = self.m_sigma_s2_gpu = np.float32(0)
        self.patch_potential = np.zeros(self.n, np.float32)

    # ---- robust statistics ---------------------------------------------------------------
    def initializeEMValues(self):                                         # :78-95
        self.scale[:] = 1.0
        self.patch_weight[:] = 1.0
        self.e.UpdateScaleVector(self._local(self.scale), self._local(self.patch_weight))
        self.e.InitializeEMValues()

    def _local(self, v):
        return np.ascontiguousarray(v[self.lo:self.hi])

    def _exchange(self, mine, pot_local=None):
        """svr::Shard::exchange (csrc/svr_shard.h): ONE sum all-reduce in which a rank fills only its own entries ->
        (all[world][len(mine)], the complete potentials or None); the scale vector rides along when it is stale."""
        W, R, n = self.comm.world, self.comm.rank, self.n
        nm = len(mine)
        v = np.zeros(nm * W + W + 3 * n, np.float64)
        v[R * nm:(R + 1) * nm] = mine
        flags = (1 if self._scale_stale else 0) | (4 if pot_local is not None else 0)
        v[nm * W + R] = flags + 1
        o = nm * W + W
        if self._scale_stale:
            v[o + self.lo:o + self.hi] = self.scale[self.lo:self.hi]
        if pot_local is not None:
            v[o + 2 * n + self.lo:o + 2 * n + self.hi] = pot_local
        v = self.comm.allreduce_sum(v)
        if not np.all(v[nm * W:nm * W + W] == flags + 1):
            raise RuntimeError("exchange: the ranks are not in the same step of the reconstruction")
        if self._scale_stale:
            self.scale = v[o:o + n].astype(np.float32)
            self._scale_stale = False
        return v[:nm * W].reshape(W, nm), (v[o + 2 * n:o + 3 * n].astype(np.float32) if pot_local is not None else None)

    def InitializeRobustStatistics(self):                                 # :793-845
        sa, sb = self.e.RobustStatisticsSums()
        if self.comm:
            allv, _ = self._exchange([sa, sb])
            sa, sb = 0.0, 0.0
            for r in range(self.comm.world):                               # rank order: the same bits on every rank
                sa += allv[r][0]
                sb += allv[r][1]
        if sb == 0:
            raise RuntimeError("ERROR: sb = 0!! no sigma computed!")      # the reference exits here
        self.m_sigma_gpu = np.float32(np.float32(sa) / np.float32(sb))
        self.m_sigma_s_gpu = np.float32(0.025)
        self.m_mix_gpu = np.float32(0.9)
        self.m_mix_s_gpu = np.float32(0.9)
        self.m_m_gpu = np.float32(np.float32(1.0) / (np.float32(2.1) * np.float32(self.m_max_intensity)
                                                    - np.float32(1.9) * np.float32(self.m_min_intensity)))

    def EStep(self):                                                      # :224-556
        pot_dev = self.e.EStep(float(self.m_m_gpu), float(self.m_sigma_gpu), float(self.m_mix_gpu))
        if self.comm:
            _, pot_dev = self._exchange([], np.asarray(pot_dev, np.float32))   # (and the scale vector)
        # the reference writes stack i's potentials to patch_potential[j], j = index within the stack,
        # without the stack offset (:256-276): later stacks overwrite the head, the tail stays 0
        pp = np.zeros(self.n, np.float32)
        ofs = 0
        for c in self.counts:
            pp[:c] = pot_dev[ofs:ofs + c]
            ofs += c
        sc, pw = self.scale, self.patch_weight.copy()
        pp[(sc < 0.2) | (sc > 5)] = -1                                    # :307-311
        valid = pp >= 0
        ppd, pwd = pp.astype(np.float64), pw.astype(np.float64)
        s, d = ((pp * pw).astype(np.float64))[valid].sum(), pwd[valid].sum()
        s2, d2 = (ppd * (1.0 - pwd))[valid].sum(), (1.0 - pwd)[valid].sum()
        maxs, mins = 0.0, 1.0
        if valid.any():
            maxs, mins = max(0.0, float(ppd[valid].max())), min(1.0, float(ppd[valid].min()))
        self.m_mean_s_gpu = np.float32(s / d) if d > 0 else np.float32(mins)
        self.m_mean_s2_gpu = np.float32(s2 / d2) if d2 > 0 else np.float32((maxs + float(self.m_mean_s_gpu)) / 2.0)
        # (pp - mean) products are float expressions widened on accumulation (:364-372)
        dv = (pp - self.m_mean_s_gpu)
        dv2 = (pp - self.m_mean_s2_gpu)
        s = ((dv * dv * pw).astype(np.float64))[valid].sum()
        d = pwd[valid].sum()
        s2 = ((dv2 * dv2 * (np.float32(1) - pw)).astype(np.float64))[valid].sum()
        d2 = (1 - pwd)[valid].sum()
        floor = float(self.m_step * self.m_step) / 6.28                  # float product over a double literal
        if s > 0 and d > 0:
            self.m_sigma_s_gpu = np.float32(s / d)
            if self.m_sigma_s_gpu < floor:
                self.m_sigma_s_gpu = np.float32(floor)
        else:
            self.m_sigma_s_gpu = np.float32(0.025)
        if s2 > 0 and d2 > 0:
            self.m_sigma_s2_gpu = np.float32(s2 / d2)
            if self.m_sigma_s2_gpu < floor:
                self.m_sigma_s2_gpu = np.float32(floor)
        else:
            dm = self.m_mean_s2_gpu - self.m_mean_s_gpu
            self.m_sigma_s2_gpu = np.float32(dm * dm / np.float32(4))
            if self.m_sigma_s2_gpu < floor:
                self.m_sigma_s2_gpu = np.float32(floor)
        for i in range(self.n):                                           # :415-452
            p = pp[i]
            if p == -1:
                pw[i] = 0
                continue
            if d <= 0 or self.m_mean_s2_gpu <= self.m_mean_s_gpu:
                pw[i] = 1
                continue
            gs1 = float(_G(p - self.m_mean_s_gpu, self.m_sigma_s_gpu)) if p < self.m_mean_s2_gpu else 0.0
            gs2 = float(_G(p - self.m_mean_s2_gpu, self.m_sigma_s2_gpu)) if p > self.m_mean_s_gpu else 0.0
            lik = gs1 * float(self.m_mix_s_gpu) + gs2 * (1 - float(self.m_mix_s_gpu))
            if lik > 0:
                pw[i] = np.float32(gs1 * float(self.m_mix_s_gpu) / lik)
            else:
                if p <= self.m_mean_s_gpu:
                    pw[i] = 1
                if p >= self.m_mean_s2_gpu:
                    pw[i] = 0
                if self.m_mean_s_gpu < p < self.m_mean_s2_gpu:
                    pw[i] = 1
        num = int(valid.sum())                                            # :455-468
        self.m_mix_s_gpu = np.float32(pw.astype(np.float64)[valid].sum() / num) if num > 0 else np.float32(0.9)
        self.patch_potential = pp
        self.patch_weight = pw
        self.e.UpdateScaleVector(self._local(self.scale), self._local(self.patch_weight))           # copyToWeightsAndScales :486-491

    def MStep(self, it):                                                  # :570-640
        s5 = [float(v) for v in self.e.MStepSums()]
        if self.comm:
            allv, _ = self._exchange(s5)
            s5 = [0.0, 0.0, 0.0, float(allv[0][3]), float(allv[0][4])]
            for r in range(self.comm.world):
                for k in range(3):
                    s5[k] += float(allv[r][k])
                s5[3], s5[4] = min(s5[3], float(allv[r][3])), max(s5[4], float(allv[r][4]))
        sigma, mix, num, mn, mx = [np.float32(v) for v in s5]
        if mix > 0:
            self.m_sigma_gpu = np.float32(sigma / mix)
        floor = np.float32(self.m_step * self.m_step) / np.float32(6.28)
        if self.m_sigma_gpu < floor:
            self.m_sigma_gpu = floor
        if it > 1:
            self.m_mix_gpu = np.float32(mix / num)
        self.m_m_gpu = np.float32(np.float32(1.0) / (mx - mn))

    def Scale(self):                                                      # :672-745
        self.scale = self.scale.copy()
        self.scale[self.lo:self.hi] = np.asarray(self.e.CalculateScaleVector(), np.float32)
        self.e.UpdateScaleVector(self._local(self.scale), self._local(self.patch_weight))           # copyToScales: no lag
        self._scale_stale = self.comm is not None                         # read next in the E-step, whose exchange completes it

    # ---- patch-to-volume registration (PBR.cpp:452-489) ----------------------------------------
    def registerPatches(self, prob):
        """PatchBased2D3DRegistration_gpu2<T>::run -- the reference's GPU variant, which its command line does not call (PBR.cpp:
        472-476 runs runHybrid, see PatchToVolumeRegistration below) -- for every stack's patches (one call: the engine holds all
        of them) against the current reconstruction, then the new transformations go back into the engine; updates
        prob.slice_t / slice_tinv."""
        t, ti, counters = self.e.register_patches(prob.patch_ri2w, prob.patch_mo, prob.patch_invmo, prob.slice_t)
        prob.slice_t, prob.slice_tinv = t, ti
        self.e.SetSliceMatrices(t, ti, prob.slice_i2w, prob.slice_w2i, prob.slice_i2w, prob.slice_w2i, prob.recon_i2w, prob.recon_w2i)
        return counters

    def PatchToVolumeRegistration(self, prob, T, recon_attr, backend=None, volume=None):
        """patchBased2D3DRegistration<T>::runHybrid (patchBased2D3DRegistration.cpp:184-225), what PBR.cpp:452-489 runs between the
        outer iterations: the IRTK slice-to-volume schedule on every patch (csrc/irtk_reg.cpp, similarities on the GPU) against
        the host copy of the reconstruction.  T: float64 [n][4][4], the registrator's own transformations; returns the new ones
        and the number of similarity evaluations, and hands them to the engine.  `volume`: an existing reconstruction target that
        stands in for the device copy before the first iteration (PBR.cpp:310-314, 456-459 upload it and read it back)."""
        from fetalreconstruction_amd import host
        vx, vy, vz = prob.vsize
        vol = np.asarray(self.e.syncCPU() if volume is None else volume, np.float32).reshape(vz, vy, vx)   # m_GPURecon.copyToHost
        hip = self.e if hasattr(self.e, "_h") else None
        T, evals = host.SliceToVolumeRegistration(hip, prob.slices, prob.slice_attr, T, recon_attr, vol, backend=backend, no_resample=True)
        t = np.stack([geo.to_matrix4(m) for m in T])
        ti = np.stack([geo.to_matrix4(np.linalg.inv(m)) for m in T])                       # updateTransformationMatrices
        prob.slice_t, prob.slice_tinv = t, ti
        self.e.SetSliceMatrices(t, ti, prob.slice_i2w, prob.slice_w2i, prob.slice_i2w, prob.slice_w2i, prob.recon_i2w, prob.recon_w2i)
        return T, evals

    # ---- the loop ------------------------------------------------------------------------
    def reconstruct_iteration(self, rec_iterations):
        """One outer iteration without the patch registration (PBR.cpp:490-548)."""
        self.initializeEMValues()
        if self.comm:
            self.e.GaussianReconstructionLocal()
            self.comm.allreduce_volume_pair(self.e, 0)
            self.e.GaussianReconstructionFinish()
        else:
            self.e.GaussianReconstruction()        # reset + patchBasedPSFReconstruction_gpu + equalize
        self.e.SimulateSlices()
        self.InitializeRobustStatistics()
        self.EStep()
        for i in range(rec_iterations):
            self.Scale()
            args = (self.m_adaptive, float(self.m_alpha), self.m_min_intensity, self.m_max_intensity, float(self.m_delta), float(self.m_lambda))
            if self.comm:
                self.e.SuperresolutionBackproject(self._local(self.patch_weight))
                self.comm.allreduce_volume_pair(self.e, 2)
                self.e.SuperresolutionUpdate(*args)
            else:
                self.e.Superresolution(i + 1, self._local(self.patch_weight), *args)   # resetAddonCmap + run + regularize
            self.e.SimulateSlices()
            self.MStep(i + 1)
            self.EStep()
