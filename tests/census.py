"""Literal-vs-canonical census of the PSF walk (test infrastructure: uses the oracle).

The device implements the oracle's CANONICAL float32 sequence bit for bit; the reference's own arithmetic is the LITERAL
one (getPSFParamsPrecomp + calcPSF op for op, RC.cu:112-174, with libm standing in for CUDA's sinf / expf).  The
epsilon-skip (RC.cu:238) is a discontinuous float compare, so which taps a pixel processes can differ between the two
by the taps whose |oldPSF - psf| lies within float round-off of 1e-5.  `census` counts exactly that on sampled pixels:
flipped keep decisions, the PSF mass they carry, and the largest value difference."""
import numpy as np


def census(prob, oracle_mod, n_pixels, seed=0, pvr=False):
    """-> dict(pixels, taps, kept_lit, kept_can, flips, flip_rate, flipped_mass_rel, max_abs_dpsf, pixels_with_flips,
    sume_rel_max).  flip = a tap processed in one mode and skipped in the other."""
    lit = oracle_mod.OracleReconstruction(prob, oracle_mod.LITERAL, pvr=pvr)
    can = oracle_mod.OracleReconstruction(prob, oracle_mod.CANON, pvr=pvr)
    act = np.argwhere(prob.slices != -1)
    rng = np.random.default_rng(seed)
    pick = act[rng.choice(len(act), min(n_pixels, len(act)), replace=False)]
    S = 12 if pvr else 16
    taps = flips = kl = kc = pw = 0
    mass_flip = mass_all = 0.0
    dmax = 0.0
    srel = 0.0
    for sl, py, px in pick:
        _, bl, vl, cl = lit.tap_census(sl, px, py, with_vals=True)
        _, bc, vc, cc = can.tap_census(sl, px, py, with_vals=True)
        assert np.array_equal(cl, cc)                                   # centre voxel: index work, identical
        keep_l = np.unpackbits(bl.view(np.uint8), bitorder="little").astype(bool)
        keep_c = np.unpackbits(bc.view(np.uint8), bitorder="little").astype(bool)
        x = keep_l != keep_c
        n = int(x.sum())
        flips += n
        pw += n > 0
        kl += int(keep_l.sum())
        kc += int(keep_c.sum())
        taps += S ** 3
        raw_l = lit.psf_values(sl, px, py)
        raw_c = can.psf_values(sl, px, py)
        ok = np.isfinite(raw_l) & np.isfinite(raw_c)
        dmax = max(dmax, float(np.abs(raw_l[ok] - raw_c[ok]).max()))
        tot_l = float(np.where(keep_l, np.nan_to_num(raw_l), 0).sum())
        tot_c = float(np.where(keep_c, np.nan_to_num(raw_c), 0).sum())
        mass_all += tot_l
        mass_flip += float(np.where(x, np.nan_to_num(np.maximum(raw_l, raw_c)), 0).sum())
        if tot_l > 0:
            srel = max(srel, abs(tot_l - tot_c) / tot_l)
    return dict(pixels=len(pick), taps=taps, kept_lit=kl, kept_can=kc, flips=flips, flip_rate=flips / max(taps, 1),
                flips_per_kept=flips / max(kl, 1), flipped_mass_rel=mass_flip / max(mass_all, 1e-30), max_abs_dpsf=dmax,
                pixels_with_flips=int(pw), sume_rel_max=srel)


def fastmath_census(prob, oracle_mod, n_pixels, seed=0, pvr=False):
    """The error envelope of the reference's own build (`--use_fast_math`, source/cmake/FindSciCuda.cmake:65-68) around the
    literal sequence, on the same sampled pixels as `census`: how many skip decisions ANY arithmetic inside the envelope may
    take differently (an upper bound), the PSF mass behind them, and where the canonical sequence lies relative to the envelope.
    -> dict(pixels, taps, uncertain_rate, pixels_uncertain, uncertain_mass_rel, sume_rel_max, env_max, env_mean, canon_outside_rate,
    canon_dmax, canon_over_env_max, canon_flips, canon_flips_explained)"""
    lit = oracle_mod.OracleReconstruction(prob, oracle_mod.LITERAL, pvr=pvr)
    act = np.argwhere(prob.slices != -1)
    rng = np.random.default_rng(seed)
    pick = act[rng.choice(len(act), min(n_pixels, len(act)), replace=False)]
    r = lit.fastmath_census(pick)
    return dict(pixels=len(pick), taps=int(r["taps"]), uncertain=int(r["uncertain"]), uncertain_rate=r["uncertain"] / max(r["taps"], 1),
                pixels_uncertain=int(r["pixels_uncertain"]), uncertain_mass_rel=r["mass_uncertain"] / max(r["mass_kept"], 1e-30),
                sume_rel_max=r["sume_rel_max"], env_max=r["env_max"], env_mean=r["env_mean"],
                canon_outside_rate=r["canon_outside"] / max(r["taps"], 1), canon_dmax=r["canon_dmax"],
                canon_over_env_max=r["canon_over_env_max"], canon_flips=int(r["canon_flips"]), canon_flips_explained=int(r["canon_flips_explained"]))
