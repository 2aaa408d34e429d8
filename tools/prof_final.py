#!/usr/bin/env python3
"""Round-6 evidence (the round-2 script, re-pointed), run on the GPU box from the repo root (one gpurun call):

  1. bench lines: P4 (with the CPU baseline) and S8
  2. rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline`      -> profiles/r06_kernel_stats_p4.txt
  3. rocprofv3 --kernel-trace --pmc passes (SQ set A, SQ set B, FETCH_SIZE, WRITE_SIZE -- each its own pass) over
     tools/run_sr_kernels.py on P4 and on S8                                -> profiles/r06_pmc_p4.txt, r06_pmc_s8.txt
  4. profiles/r06_traffic.json: FETCH_SIZE / WRITE_SIZE of the scatter and the gather per launch (KB -> bytes), which
     bench.py reads for roofline.traffic; then the P4 bench line again with it           -> profiles/r06_bench_p4.json
"""
import csv
import glob
import json
import os
import re
import sqlite3
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r06"
OUT = os.path.join(R, "gpurun_out", TAG)
PROF = os.path.join(R, "profiles")
ENV = dict(os.environ, TMPDIR="/tmp")


def sh(cmd, log=None, cwd="/tmp"):
    p = subprocess.run(cmd, cwd=cwd, env=ENV, capture_output=True, text=True)
    if log:
        open(log, "w").write(p.stdout + "\n--- stderr ---\n" + p.stderr[-6000:])
    return p


def last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def pmc(workload, tag, opts=()):
    d = os.path.join(OUT, "pmc_" + tag)
    p = sh([sys.executable, os.path.join(R, "tools", "pmc.py"), d, "--filter", "k", "--", sys.executable,
            os.path.join(R, "tools", "run_sr_kernels.py"), workload, *opts])
    txt = p.stdout
    head = (f"# rocprofv3 PMC counters of the SR kernels on {workload} {' '.join(opts)} (1 GPU), round 6.  tools/pmc.py over tools/run_sr_kernels.py: one\n"
            "# rocprofv3 --kernel-trace --pmc <set> pass per counter set (SQ set A, SQ set B, FETCH_SIZE, WRITE_SIZE), nothing else in\n"
            "# the pass.  Per kernel: mean over the last half of its dispatches.  SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are in\n"
            "# quad-cycles summed over the SIMDs, FETCH_SIZE / WRITE_SIZE in KB (uncalibrated for this access pattern: narrow LDS-staged\n"
            "# reads and float atomics -- MI355X_MICROARCH.md calibrates only wide streaming reads -- so they are reported as counted).\n")
    open(os.path.join(PROF, f"r06_pmc_{tag}.txt"), "w").write(head + txt)
    return parse_pmc(txt)


def parse_pmc(txt):
    vals = {}
    cur = None
    for line in txt.splitlines():
        m = re.match(r"^(\S.*?)\s+dispatches/pass", line)
        if m:
            cur = m.group(1)
            vals[cur] = {}
        elif cur and re.match(r"^\s+[A-Z_]+\s", line):
            k, v = line.split()[:2]
            vals[cur][k] = float(v)
    return vals


def stats_only():
    """2. alone: the kernel trace of the bench command (headline workload only: `--no-s8`, so that the per-kernel averages are P4's)"""
    os.makedirs(OUT, exist_ok=True)
    bench = os.path.join(R, "bench.py")
    j4 = json.load(open(os.path.join(PROF, "r06_bench_p4.json"))) if os.path.exists(os.path.join(PROF, "r06_bench_p4.json")) else None
    j8 = json.load(open(os.path.join(PROF, "r06_bench_s8.json"))) if os.path.exists(os.path.join(PROF, "r06_bench_s8.json")) else None
    kernel_stats(bench, j4, j8)


def build_traffic(get):
    """profiles/r06_traffic.json from the PMC passes (get(workload, tag, opts) -> {kernel: {counter: value}}: a fresh rocprofv3 run, or a committed
    profiles/r06_pmc_<tag>.txt read back: --traffic-only)"""
    traffic = {"source": "profiles/r06_pmc_<workload>.txt (rocprofv3 --pmc SQ_INSTS_VALU / FETCH_SIZE / WRITE_SIZE, separate passes, KB -> bytes; "
                         "per pass the kernels of one scatter / gather launch summed)"}

    def collect(vals, pats):
        """the kernels of one pass; the FIRST pattern names its main kernel and must be there"""
        ks = [n for n in vals if any(p_ in n for p_ in pats)]
        if not ks or not any(pats[0] in n for n in ks) or not all("FETCH_SIZE" in vals[k] and "WRITE_SIZE" in vals[k] for k in ks):
            return None
        return {"kernels": ks, "fetch_bytes": sum(vals[k]["FETCH_SIZE"] for k in ks) * 1024.0, "write_bytes": sum(vals[k]["WRITE_SIZE"] for k in ks) * 1024.0,
                "valu_insts": sum(vals[k].get("SQ_INSTS_VALU", 0.0) for k in ks)}

    # slice-to-volume workloads: the default (coefficient table: the passes that stream it, and the gather that writes it) and coeff_table=0 (every
    # tap evaluated); patch-based workloads: their default (evaluated)
    todo = [("P4", "p4", ()), ("P4", "p4_on_the_fly", ("coeff_table=0",)), ("S8", "s8", ()), ("S8", "s8_on_the_fly", ("coeff_table=0",)),
            ("PVR4", "pvr4", ()), ("PVR8spx", "pvr8spx", ())]
    for wl, tag, opts in todo:
        v = get(wl, tag, opts)
        if not v:
            continue
        pv = wl.startswith("PVR")
        nsup, isp = ("12", "true") if pv else ("16", "false")
        e = traffic.setdefault(wl, {})
        if pv or opts:
            e["back"] = collect(v, ("back_cell_kernel<%s, %s, 0>" % (nsup, isp), "k_cell_combine", "k_cell_factors")) or collect(v, ("back_wave_kernel<%s, %s, false>" % (nsup, isp),))
            e["forward"] = collect(v, ("fwd_cell_kernel<%s, %s, 0, false>" % (nsup, isp), "k_cell_gather_finish", "k_cell_gfactors")) or collect(v, ("fwd_unit_kernel<false, %s, %s, false>" % (nsup, isp),))
        else:
            e["back_table"] = collect(v, ("back_cell_kernel<16, false, 1>", "k_cell_combine", "k_cell_factors"))
            e["forward_table"] = collect(v, ("fwd_cell_kernel<16, false, 2, false>", "k_cell_gather_finish")) or collect(v, ("fwd_unit_kernel<false, 16, false, true>",))
            e["forward_store"] = collect(v, ("fwd_cell_kernel<16, false, 3, false>", "k_cell_gather_finish"))
            e["back_store"] = collect(v, ("back_cell_kernel<16, false, 3>", "k_cell_combine", "k_cell_factors"))
        e["update"] = e.get("update") or collect(v, ("k_regul_fused",))
        traffic[wl] = {k: x for k, x in e.items() if x}
    if traffic.get("P4", {}).get("back_table") or traffic.get("P4", {}).get("back"):
        json.dump(traffic, open(os.path.join(PROF, "r06_traffic.json"), "w"), indent=1)
    return traffic


def main():
    if "--stats-only" in sys.argv:
        return stats_only()
    if "--traffic-only" in sys.argv:                       # from the committed PMC summaries (no GPU)
        def read_back(wl, tag, opts):
            f = os.path.join(PROF, f"r06_pmc_{tag}.txt")
            return parse_pmc(open(f).read()) if os.path.exists(f) else None
        print(json.dumps(build_traffic(read_back).get("S8")))
        return
    if "--timeline-only" in sys.argv:
        os.makedirs(OUT, exist_ok=True)
        return timeline(os.path.join(R, "bench.py"))
    os.makedirs(OUT, exist_ok=True)
    bench = os.path.join(R, "bench.py")
    # 3. PMC first (the traffic file must exist before the final bench line)
    traffic = build_traffic(lambda wl, tag, opts: pmc(wl, tag, opts))
    # 1. bench lines
    b4 = sh([sys.executable, bench], os.path.join(OUT, "bench_p4.log"), cwd=R)
    j4 = last_json(b4.stdout)
    if j4:
        json.dump(j4, open(os.path.join(PROF, "r06_bench_p4.json"), "w"), indent=1)
    b8 = sh([sys.executable, bench, "--workload", "S8", "--no-cpu-baseline"], os.path.join(OUT, "bench_s8.log"), cwd=R)
    j8 = last_json(b8.stdout)
    if j8:
        json.dump(j8, open(os.path.join(PROF, "r06_bench_s8.json"), "w"), indent=1)
    for wl in ("PVR4", "PVR8spx"):
        bp = sh([sys.executable, bench, "--workload", wl, "--no-cpu-baseline"], os.path.join(OUT, f"bench_{wl.lower()}.log"), cwd=R)
        jp = last_json(bp.stdout)
        if jp:
            json.dump(jp, open(os.path.join(PROF, f"r06_bench_{wl.lower()}.json"), "w"), indent=1)
    kernel_stats(bench, j4, j8)
    timeline(bench)
    print(json.dumps(traffic.get("P4")))



def timeline(bench):
    """5. every dispatch of one outer iteration of the bench's schedule on P4: three steps that stream the table, one whose scatter evaluates and whose
    gather writes it (profiles/r06_step_timeline_p4.txt)"""
    import shutil
    d = os.path.join(OUT, "tl")
    shutil.rmtree(d, ignore_errors=True)
    sh(["timeout", "420", "rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable, bench, "--steps", "8", "--warmup", "4",
        "--no-cpu-baseline", "--no-s8", "--no-coeff-table"], os.path.join(OUT, "tl.log"))
    p = subprocess.run([sys.executable, os.path.join(R, "tools", "step_timeline.py"), d, "4"], capture_output=True, text=True, env=dict(ENV, STEP_FROM="5"))
    head = ("# round 6, final binary: cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-s8 --no-coeff-table;\n"
            "# STEP_FROM=5 python tools/step_timeline.py DIR 4   (P4, one MI355X, the default mode: the first four steps of the first timed pass -- kernel timers OFF -- i.e. one outer\n"
            "# iteration of the schedule: its first step follows InitializeEMValues / InitializeRobustStatistics / EStep and the throwing away of the coefficient table,\n"
            "# its scatter evaluates and writes the table; every other pass streams it)\n")
    open(os.path.join(PROF, "r06_step_timeline_p4.txt"), "w").write(head + p.stdout)
    print(p.stdout[:1500])


def kernel_stats(bench, j4, j8):
    # 2. kernel trace of the bench command
    import shutil
    d = os.path.join(OUT, "stats")
    shutil.rmtree(d, ignore_errors=True)
    p = sh(["timeout", "420", "rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "bench", "--", sys.executable, bench, "--no-cpu-baseline", "--no-s8"],   # (timeout: rocprofv3 has been seen to hang at exit after writing its database)
           os.path.join(OUT, "stats.log"))
    jt = last_json(p.stdout)
    lines = ["# round 6: cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d gpurun_out/r06/stats -o bench -- python bench.py --no-cpu-baseline --no-s8",
             "# (tools/prof_final.py; rocprofv3 of this image writes a rocpd database: `top_kernels` view below.  --no-s8: the default command also measures S8",
             "#  after the headline workload in the same launch -- with it the per-kernel averages below would mix P4's 3 ms launches with S8's 33 ms ones)"]
    if jt:
        lines.append("# bench line of the profiled run: value %.2f MVoxels/s, %.3f ms/step; scatter avg launch %.3f ms, gather %.3f ms by HIP events; roofline.frac %.4f"
                     % (jt["value"], jt["ms_per_step"], jt["kernel_ms"]["backproject"], jt["kernel_ms"]["forward"], jt["roofline"]["frac"]))
    if j4:
        lines.append("# unprofiled P4 line of the same call: value %.2f MVoxels/s, %.3f ms/step, scatter %.3f ms, gather %.3f ms, cpu_baseline %s"
                     % (j4["value"], j4["ms_per_step"], j4["kernel_ms"]["backproject"], j4["kernel_ms"]["forward"], json.dumps(j4.get("cpu_baseline"))))
    if j8:
        lines.append("# unprofiled S8 line: value %.2f MVoxels/s, %.3f ms/step, scatter %.3f ms, gather %.3f ms"
                     % (j8["value"], j8["ms_per_step"], j8["kernel_ms"]["backproject"], j8["kernel_ms"]["forward"]))
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if dbs:
        c = sqlite3.connect(dbs[0])
        try:
            rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
            lines.append("%7s %14s %12s %7s  %s" % ("calls", "total_us", "avg_us", "pct", "kernel"))
            lines += ["%7d %14.1f %12.1f %7.2f  %s" % (k, t, a, pc, short(n)) for n, k, t, a, pc in rows[:24]]
            # the on-the-fly instantiations; the bench times its K steps on the fly first, then the same K with the table
            for pat in ("back_cell_kernel<16, false, 1>", "fwd_cell_kernel<16, false, 2, false>", "back_cell_kernel<16, false, 3>", "fwd_cell_kernel<16, false, 3, false>", "back_cell_kernel<16, false, 0>", "fwd_cell_kernel<16, false, 0, false>"):
                dd = [r[0] / 1e3 for r in c.execute("select (end - start) from kernels where name like ? order by start", ("%" + pat + "%",)).fetchall()]
                if dd and jt:
                    k = jt["steps"]
                    w = jt["warmup"]
                    lines.append("# %s dispatches in order, us: %s" % (pat, " ".join("%.0f" % v for v in dd)))
                    lines.append("#   %d dispatches, average %.1f us (kernel trace; the bench line's HIP-event figures per kind of pass: roofline.*.avg_launch_ms)" % (len(dd), sum(dd) / max(len(dd), 1)))
        except Exception as ex:
            lines.append("# could not read the rocpd database: %r" % (ex,))
    else:
        csvs = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if csvs:
            lines += open(csvs[0]).read().splitlines()[:30]
    open(os.path.join(PROF, "r06_kernel_stats_p4.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main()
