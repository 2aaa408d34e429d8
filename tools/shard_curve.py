#!/usr/bin/env python3
"""profiles/r06_shard_projection.json: what bench.py --gpus N is PROJECTED to read on P4 and S8 at N = 2, 4, 8, in both modes (every tap evaluated /
the coefficient table in its steady passes), from per-rank kernel times measured rank after rank on ONE GPU (tools/shard_probe.py) and stated
link rates.  No collective runs here; bench.py's N > 1 line prints `speedup_vs_projection` against these steps, so that the first line from real
hardware explains itself.  usage (GPU box, repo root): python tools/shard_curve.py [--out profiles/r06_shard_projection.json] [P4 S8] [--worlds 2 4 8]"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tools"))
import shard_probe  # noqa: E402


def main():
    argv = sys.argv[1:]
    out = os.path.join(R, "profiles", "r06_shard_projection.json")
    if "--out" in argv:
        out = argv[argv.index("--out") + 1]
    worlds = [2, 4, 8]
    if "--worlds" in argv:
        i = argv.index("--worlds") + 1
        worlds = []
        while i < len(argv) and argv[i].isdigit():
            worlds.append(int(argv[i]))
            i += 1
    wls = [a for a in argv if a in ("P4", "S8", "PVR4", "PVR8spx", "tiny")] or ["P4", "S8"]
    res = {"note": "PROJECTIONS from one-GPU per-rank kernel times (tools/shard_probe.py via tools/shard_curve.py): every rank of an N-rank run measured alone on one MI355X, "
                   "the collectives priced at stated xGMI rates (projection.at_link_rates).  Nothing here ran on two GPUs.  `table`: the steady passes of the "
                   "default mode (the step of an outer iteration that rewrites the coefficient table is not in it); `on_the_fly`: svr_set_option(coeff_table, 0).",
           "curves": {}}
    for wl in wls:
        res["curves"][wl] = {}
        for n in worlds:
            e = {}
            for mode, opt in (("table", 1), ("on_the_fly", 0)):
                if wl.startswith("PVR") and mode == "table":
                    continue
                t0 = time.time()
                r = shard_probe.run(wl, n, reps=4, opts=[("coeff_table", opt)] if not wl.startswith("PVR") else [])
                p = r["projection"]
                e[mode] = {"one_gpu_kernels_ms": p["one_gpu_kernels_ms"], "max_rank_psf_em_ms": p["max_rank_psf_em_ms"], "shard_overhead": p["shard_overhead"],
                           "slab": p["slab"], "replicated": p["replicated"], "at_link_rates": p["at_link_rates"], "assumptions": p["assumptions"], "label": p["label"],
                           "ranks_backproject_ms": [k["backproject"] for k in r["shards"]], "ranks_forward_ms": [k["forward"] for k in r["shards"]],
                           "wall_s": round(time.time() - t0, 1)}
                print(f"[{wl} x{n} {mode}] projected step {p['slab']['step_ms']:.3f} ms = {p['slab']['speedup']:.2f} x one GPU's {p['one_gpu_kernels_ms']:.3f} ms "
                      f"(slab update, 7 x 76.8 x 0.5 GB/s); {time.time() - t0:.0f} s", flush=True)
            res["curves"][wl][str(n)] = e
            json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
