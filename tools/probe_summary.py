#!/usr/bin/env python3
"""print what tools/shard_probe.py measured: usage: probe_summary.py FILE.json ..."""
import json
import sys

for f in sys.argv[1:]:
    try:
        r = json.load(open(f))
    except Exception as e:          # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    p, sh = r.get("projection", {}), r["shards"]
    print(r["workload"], r.get("layout"), "world", r["world"], "wall", r.get("wall_s"), "overhead %.3f" % p.get("shard_overhead", 0), "slab %.2fx" % p.get("slab", {}).get("speedup", 0),
          "max psf+em %.3f" % p.get("max_rank_psf_em_ms", 0), "one gpu %.3f" % p.get("one_gpu_kernels_ms", 0), "| full back %.3f fwd %.3f" % (r["full"]["backproject"], r["full"]["forward"]))
    print("   back", [round(k["backproject"], 3) for k in sh], "sum %.2f" % sum(k["backproject"] for k in sh))
    print("   fwd ", [round(k["forward"], 3) for k in sh], "sum %.2f" % sum(k["forward"] for k in sh))
    print("   items", [k["cells"]["items"] for k in sh], "of", r["full"]["cells"]["items"], "| staging MB", [k["cells"]["staging_bytes"] >> 20 for k in sh])
    for name, v in p.get("at_link_rates", {}).items():
        print("   link %-50s slab %.2fx (step %.2f ms, collectives %.2f ms)   replicated %.2fx" % (name, v["speedup_slab"], v["step_slab_ms"], v["collectives_slab_ms"], v["speedup_replicated"]))
