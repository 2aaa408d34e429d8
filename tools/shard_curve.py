#!/usr/bin/env python3
"""profiles/r04_shard_projection_curve.json: `bench.py --workload W --shard all/N` for N = 2, 4, 8 on the one GPU of the box -- the
PROJECTED strong-scaling curve of S8 and P4 (tools/shard_probe.py: per-rank kernel times measured rank by rank; no collective runs)."""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"note": "PROJECTIONS from one-GPU per-rank kernel times (tools/shard_probe.py); nothing here ran on two GPUs", "curves": {}}
for wl in sys.argv[1:] or ["S8", "P4"]:
    out["curves"][wl] = {}
    for n in (2, 4, 8):
        p = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--workload", wl, "--shard", f"all/{n}"], cwd=R, capture_output=True, text=True)
        for line in reversed(p.stdout.strip().splitlines()):
            if line.startswith("{"):
                d = json.loads(line)
                pr = d["projection"]
                out["curves"][wl][str(n)] = {"one_gpu_kernels_ms": pr["one_gpu_kernels_ms"], "shard_overhead": pr["shard_overhead"], "max_rank_psf_em_ms": pr["max_rank_psf_em_ms"],
                                             "slab": pr["slab"], "replicated": pr["replicated"], "assumptions": pr["assumptions"], "label": pr["label"],
                                             "ranks_backproject_ms": [s["backproject"] for s in d["shards"]], "ranks_forward_ms": [s["forward"] for s in d["shards"]]}
                print(wl, n, "slab %.2fx replicated %.2fx overhead %.3f" % (pr["slab"]["speedup"], pr["replicated"]["speedup"], pr["shard_overhead"]), flush=True)
                break
json.dump(out, open(os.path.join(R, "profiles", "r04_shard_projection_curve.json"), "w"), indent=1)
