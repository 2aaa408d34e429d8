"""Superpixel patches of the patch-based path: SLICO segmentation of every slice of a stack and the 64x64
patches cut around the superpixels, with the `spxMask` the engine consumes (svr_set_spx_masks; ImagePatch2D.cuh:51).

Mirrors, host side like the reference:
  runStackSLIC<T>::segmentSLIC, rgbtolab, getLABXYSeeds, PerformSuperpixelSLICO, EnforceSuperpixelConnectivity
      source/reconstructionGPU2/runStackSLIC.cpp:55-151, 291-537, 665-840 (SLICO itself is Achanta et al.'s published code)
  PatchBasedObject<T>::generate2DSuperpixelPatches, dilatePatch
      source/reconstructionGPU2/include/patchBasedObject.cuh:347-367, 433-802
Kept quirks: segmentSLIC copies the slice into its buffer column by column (x outer, y inner) and runs SLIC on it as a
`width = ny` by `height = nx` image, i.e. on the transposed slice, and reads the labels back the same way; the
superpixel loop stops before the largest label (`idxLbl < int(maxLbl)`); patches are always 64x64 (clamped to the slice);
a patch pixel that keeps the dilated mask value but falls outside the mask image keeps the value 1.
"""
from __future__ import annotations

import copy

import numpy as np

from fetalreconstruction_amd import geometry as geo
from .registration import irtk_round


def rgbtolab(r, g, b):
    """runStackSLIC.cpp:55-110: sRGB (0..255 integers) -> CIE LAB, vectorised"""
    def lin(c):
        c = c / 255.0
        return np.where(c <= 0.04045, c / 12.92, np.power((c + 0.055) / 1.055, 2.4))
    R, G, B = lin(r.astype(np.float64)), lin(g.astype(np.float64)), lin(b.astype(np.float64))
    X = R * 0.4124564 + G * 0.3575761 + B * 0.1804375
    Y = R * 0.2126729 + G * 0.7151522 + B * 0.0721750
    Z = R * 0.0193339 + G * 0.1191920 + B * 0.9503041
    eps, kappa = 0.008856, 903.3

    def f(t):
        return np.where(t > eps, np.power(t, 1.0 / 3.0), (kappa * t + 16.0) / 116.0)
    fx, fy, fz = f(X / 0.950456), f(Y / 1.0), f(Z / 1.088754)
    return 116.0 * fy - 16.0, 500.0 * (fx - fy), 200.0 * (fy - fz)


def get_seeds(step, width, height):
    """getLABXYSeeds, runStackSLIC.cpp:111-151 -> flat indices (y * width + x) of the grid seeds"""
    xstrips = int(0.5 + width / step)
    ystrips = int(0.5 + height / step)
    xerr = width - step * xstrips
    if xerr < 0:
        xstrips -= 1
        xerr = width - step * xstrips
    yerr = height - step * ystrips
    if yerr < 0:
        ystrips -= 1
        yerr = height - step * ystrips
    xeps, yeps = xerr / xstrips, yerr / ystrips
    off = step // 2
    seeds = []
    for y in range(ystrips):
        ye = int(y * yeps)
        for x in range(xstrips):
            xe = int(x * xeps)
            seeds.append((y * step + off + ye) * width + (x * step + off + xe))
    return np.array(seeds, np.int64)


def slico(l, a, b, seeds, width, height, step):
    """PerformSuperpixelSLICO, runStackSLIC.cpp:291-437: 10 iterations of the zero-parameter SLIC.  l, a, b: flat float64
    [height * width].  The seed loop is sequential like the reference: a pixel's colour distance `distlab` is whatever the
    LAST seed whose window covers it computed, not the winner's."""
    sz = width * height
    kx, ky = (seeds % width).astype(np.float64), (seeds // width).astype(np.float64)
    kl, ka, kb = l[seeds].copy(), a[seeds].copy(), b[seeds].copy()
    numk = len(seeds)
    klabels = np.full(sz, -1, np.int64)
    distlab = np.full(sz, np.finfo(np.float64).max)
    maxlab = np.full(numk, 100.0)
    invxywt = 1.0 / (step * step)
    L, A, B = l.reshape(height, width), a.reshape(height, width), b.reshape(height, width)
    lab2 = klabels.reshape(height, width)
    dl2 = distlab.reshape(height, width)
    rows, cols = np.divmod(np.arange(sz), width)
    for itr in range(10):
        distvec = np.full((height, width), np.finfo(np.float64).max)
        for n in range(numk):
            x1, y1 = max(int(kx[n] - step), 0), max(int(ky[n] - step), 0)           # int(): truncation of the double, like the C assignment
            x2, y2 = min(int(kx[n] + step), width), min(int(ky[n] + step), height)
            if x1 >= x2 or y1 >= y2:
                continue
            yy, xx = np.mgrid[y1:y2, x1:x2]
            dlab = ((L[y1:y2, x1:x2] - kl[n]) * (L[y1:y2, x1:x2] - kl[n]) + (A[y1:y2, x1:x2] - ka[n]) * (A[y1:y2, x1:x2] - ka[n]) +
                    (B[y1:y2, x1:x2] - kb[n]) * (B[y1:y2, x1:x2] - kb[n]))
            dl2[y1:y2, x1:x2] = dlab
            dxy = (xx - kx[n]) * (xx - kx[n]) + (yy - ky[n]) * (yy - ky[n])
            dist = dlab / maxlab[n] + dxy * invxywt
            win = dist < distvec[y1:y2, x1:x2]
            distvec[y1:y2, x1:x2] = np.where(win, dist, distvec[y1:y2, x1:x2])
            lab2[y1:y2, x1:x2] = np.where(win, n, lab2[y1:y2, x1:x2])
        if itr == 0:
            maxlab[:] = 1.0
        ok = klabels >= 0
        np.maximum.at(maxlab, klabels[ok], distlab[ok])
        size = np.bincount(klabels[ok], minlength=numk).astype(np.float64)
        inv = 1.0 / np.where(size <= 0, 1.0, size)

        def mean(v):                                       # sums in raster order, like the reference's loop
            return np.bincount(klabels[ok], weights=v[ok], minlength=numk) * inv
        kl, ka, kb = mean(l), mean(a), mean(b)
        kx, ky = mean(cols.astype(np.float64)), mean(rows.astype(np.float64))
    return klabels


def enforce_connectivity(labels, width, height, num_superpixels):
    """EnforceSuperpixelConnectivity, runStackSLIC.cpp:440-537: relabel 4-connected segments in raster order; a segment of at
    most SUPSZ/4 pixels takes the label of a previously labelled neighbour of its first pixel."""
    sz = width * height
    supsz = sz // max(num_superpixels, 1)
    nl = np.full(sz, -1, np.int64)
    dx4, dy4 = (-1, 0, 1, 0), (0, -1, 0, 1)
    label, adjlabel = 0, 0
    for oindex in range(sz):
        if nl[oindex] >= 0:
            continue
        j, k = divmod(oindex, width)
        nl[oindex] = label
        for n in range(4):
            x, y = k + dx4[n], j + dy4[n]
            if 0 <= x < width and 0 <= y < height and nl[y * width + x] >= 0:
                adjlabel = nl[y * width + x]
        xs, ys = [k], [j]
        c = 0
        want = labels[oindex]
        while c < len(xs):
            for n in range(4):
                x, y = xs[c] + dx4[n], ys[c] + dy4[n]
                if 0 <= x < width and 0 <= y < height:
                    ni = y * width + x
                    if nl[ni] < 0 and labels[ni] == want:
                        xs.append(x)
                        ys.append(y)
                        nl[ni] = label
            c += 1
        if len(xs) <= supsz >> 2:
            nl[np.array(ys) * width + np.array(xs)] = adjlabel
            label -= 1
        label += 1
    return nl, label


def segment_slic(stack, spx_size):
    """runStackSLIC<T>::segmentSLIC (:665-840): the label image of every slice of `stack` [nz][ny][nx] -> float32 [nz][ny][nx]."""
    nz, ny, nx = stack.shape
    vmin, vmax = float(stack.min()), float(stack.max())
    width, height = ny, nx                                  # :704-705 -- the buffer is filled x outer, y inner
    sz = width * height
    nsp = int(sz / (spx_size[0] * spx_size[1]))
    out = np.zeros(stack.shape, np.float32)
    for z in range(nz):
        buf = stack[z].astype(np.float32).T.reshape(-1)     # p = x * ny + y
        # `(int) 255 * (v - min) / (max - min)`: the cast binds to 255, the arithmetic is float, the assignment truncates
        grey = ((np.float32(255) * (buf - np.float32(vmin))) / (np.float32(vmax) - np.float32(vmin))).astype(np.int64) if vmax > vmin \
            else np.zeros(sz, np.int64)
        l, a, b = rgbtolab(grey, grey, grey)
        step = int(np.sqrt(sz / nsp) + 0.5)
        seeds = get_seeds(step, width, height)
        kl = slico(l, a, b, seeds, width, height, step)
        cl, _ = enforce_connectivity(kl, width, height, nsp)
        out[z] = cl.reshape(nx, ny).T                       # stack_spx(x, y, z) = clabels[x * ny + y]
    return out


def dilate_patch(m):
    """dilatePatch, patchBasedObject.cuh:347-367: one 4-neighbour dilation of the pixels equal to 1"""
    one = m == 1
    grow = np.zeros_like(one)
    grow[:, :-1] |= one[:, 1:]
    grow[:, 1:] |= one[:, :-1]
    grow[:-1, :] |= one[1:, :]
    grow[1:, :] |= one[:-1, :]
    out = m.copy()
    out[grow & (m == 0)] = 1
    return out


def generate2DSuperpixelPatches(stack, mask, mask_attr, spx_size, extend_percent):
    """PatchBasedObject<T>::generate2DSuperpixelPatches (:433-802) for one pvr.Stack.
    Returns (patches float32 [n][pY][pX] with -1 outside the dilated superpixels, I2W [n][16], W2I [n][16], spxMask uint8
    [n][4096] of '1' / 0 in the 64-wide wire format, origins float64 [n][3], patch attributes)."""
    a = stack.attr
    data = np.asarray(stack.data, np.float32)
    sx, sy = int(spx_size[0]), int(spx_size[1])
    if sx > a.nx:
        sx = a.nx // 2
    if sy > a.ny:
        sy = a.ny // 2
    labels = segment_slic(data, (sx, sy))
    ratio = np.float32(extend_percent) / np.float32(100.0)
    px, py = min(64, a.nx), min(64, a.ny)
    m_w2i = geo.world_to_image(mask_attr)
    mz, my, mx = mask.shape
    patches, i2ws, w2is, masks, origins, attrs = [], [], [], [], [], []
    jj, ii = np.meshgrid(np.arange(py), np.arange(px), indexing="ij")
    pix = np.stack([ii, jj, np.zeros_like(ii), np.ones_like(ii)], -1).astype(np.float64)
    rnd = np.vectorize(irtk_round)
    for z in range(a.nz):
        lab = labels[z]
        sl_attr = copy.copy(a)
        sl_attr.nz = 1
        sl_attr.dz = stack.thickness * 2
        sl_attr.origin = geo.region_origin(a, 0, 0, z, sl_attr)
        sl_i2w, sl_w2i = geo.image_to_world(sl_attr), geo.world_to_image(sl_attr)
        for idx in range(int(lab.min()), int(lab.max())):          # `idxLbl < int(maxLbl)`: the last label is never cut out
            ys, xs = np.nonzero(lab.astype(np.int64) == idx)
            if len(xs) == 0:
                continue
            x_min, x_max, y_min, y_max = int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())
            wx, wy = x_max - x_min, y_max - y_min
            diter = int(ratio * np.float32(wx if wx > wy else wy))
            ex, ey = irtk_round((float(px) - float(wx)) / 2.0), irtk_round((float(py) - float(wy)) / 2.0)
            if x_min - ex < 0:
                x_min, x_max = 0, px
            elif x_max + ex > a.nx:
                x_max = a.nx
                x_min = x_max - px
            else:
                x_min -= ex
                x_max = x_min + px
            if y_min - ey < 0:
                y_min, y_max = 0, py
            elif y_max + ey > a.ny:
                y_max = a.ny
                y_min = y_max - py
            else:
                y_min -= ey
                y_max = y_min + py
            # patch = GetRegion(xMin, yMin, z, xMax, yMax, z + 1) with the slice's pixel size
            pa = copy.copy(sl_attr)
            pa.nx, pa.ny = px, py
            pa.origin = geo.region_origin(a, x_min, y_min, z, pa)
            p_i2w = geo.image_to_world(pa)
            w = geo.apply_points(p_i2w, pix)
            q = geo.apply_points(sl_w2i, w)
            qx, qy = rnd(q[..., 0]), rnd(q[..., 1])
            qm = geo.apply_points(m_w2i, w)
            m1, m2, m3 = rnd(qm[..., 0]), rnd(qm[..., 1]), rnd(qm[..., 2])
            in_slice = (qx >= 0) & (qy >= 0) & (qx < a.nx) & (qy < a.ny)
            in_maskimg = (m1 >= 0) & (m2 >= 0) & (m3 >= 0) & (m1 < mx) & (m2 < my) & (m3 < mz)
            mval = mask[np.clip(m3, 0, mz - 1), np.clip(m2, 0, my - 1), np.clip(m1, 0, mx - 1)]
            lv = lab[np.clip(qy, 0, a.ny - 1), np.clip(qx, 0, a.nx - 1)]
            pm = np.where(in_slice & in_maskimg & (mval > 0) & (lv.astype(np.int64) == idx), 1.0, 0.0)
            count = int((pm > 0).sum())
            if count < 2 or count < np.float32(1.0) / np.float32(4.0) * np.float32(sy) * np.float32(sx):
                continue
            for _ in range(diter):
                pm = dilate_patch(pm)
            sv = data[z][np.clip(qy, 0, a.ny - 1), np.clip(qx, 0, a.nx - 1)]
            val = np.where(pm == 0, -1.0, np.where(in_slice & in_maskimg, np.where(mval > 0, sv, -1.0), pm))
            msk = np.zeros((64, 64), np.uint8)
            msk[:py, :px] = np.where(val != -1, ord("1"), 0)
            patches.append(val.astype(np.float32))
            i2ws.append(geo.to_matrix4(p_i2w))
            w2is.append(geo.to_matrix4(geo.world_to_image(pa)))
            masks.append(msk.reshape(-1))
            origins.append(np.asarray(pa.origin, np.float64).copy())
            attrs.append(pa)
    n = len(patches)
    return (np.stack(patches) if n else np.zeros((0, py, px), np.float32), np.stack(i2ws) if n else np.zeros((0, 16), np.float32),
            np.stack(w2is) if n else np.zeros((0, 16), np.float32), np.stack(masks) if n else np.zeros((0, 4096), np.uint8),
            np.stack(origins) if n else np.zeros((0, 3)), attrs)
