#!/usr/bin/env python3
"""profiles/r05_shard_projection.json: `bench.py --workload W --shard all/8` for the four bench workloads on the one GPU of the
box (tools/shard_probe.py: per-rank kernel times measured one rank after the other + the PROJECTED step; no collective runs)."""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"note": "bench.py --workload W --shard all/8 on ONE MI355X (tools/shard_probe.py): per-rank kernel times of an 8-rank run measured one rank "
               "after the other on one GPU, and the PROJECTED step (no collective was run; the link rates are stated assumptions: projection.at_link_rates "
               "prices the collectives at 269, 153 and 76.8 GB/s per direction).  Round 5: the units are dealt spatially (the r-th eighth of every stack per "
               "rank, sharding.shard_units), the combine reads one item_of word per wavefront, and a repetition's state is restored device to device.",
       "workloads": {}}
for wl in sys.argv[1:] or ["S8", "P4", "PVR8spx", "PVR4"]:
    p = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--workload", wl, "--shard", "all/8", "--steps", "24"],   # (12 timed repetitions per rank: with 5, one slow repetition moved a rank by 4 %)
                       cwd=R, capture_output=True, text=True)
    for line in reversed(p.stdout.strip().splitlines()):
        if line.startswith("{"):
            out["workloads"][wl] = json.loads(line)
            break
    else:
        print(wl, "failed", p.stderr[-2000:], file=sys.stderr)
json.dump(out, open(os.path.join(R, "profiles", "r05_shard_projection.json"), "w"), indent=1)
for wl, v in out["workloads"].items():
    pr = v.get("projection", {})
    print(wl, "layout", v.get("layout"), "overhead %.3f" % pr.get("shard_overhead", 0), "replicated %.2fx" % pr.get("replicated", {}).get("speedup", 0), "slab %.2fx" % pr.get("slab", {}).get("speedup", 0),
          "| slab at the three link rates:", " ".join("%.2fx" % x["speedup_slab"] for x in pr.get("at_link_rates", {}).values()))
