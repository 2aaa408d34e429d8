"""profiles/r01_g_final_kernel_stats.txt from the outputs of tools/prof_final.sh (gpurun_out/prof_g/bench_results.db = rocprofv3's rocpd
database of `bench.py --no-cpu-baseline`, plus the bench lines of the same call)."""
import json, sqlite3, sys
R = '/root/repo'
c = sqlite3.connect(f'{R}/gpurun_out/prof_g/bench_results.db')
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
last = lambda p: json.loads([l for l in open(p).read().splitlines() if l.startswith('{"metric"')][-1])
line, b4, s8 = last(f'{R}/gpurun_out/bench_g.log'), last(f'{R}/gpurun_out/bench_p4.json'), last(f'{R}/gpurun_out/bench_s8.json')
out = ["# round 1, final state of the tree",
       "# cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d gpurun_out/prof_g -o bench -- python bench.py --no-cpu-baseline",
       "# (tools/prof_final.sh; rocprofv3 of this image writes a rocpd database, the table below is its `top_kernels` view; tools/prof_summary.py)",
       "# bench line of the profiled run: value %.2f MVoxels/s, %.3f ms/step, back_plane_kernel avg launch %.3f ms by HIP events, roofline.frac %.4f" % (
           line["value"], line["ms_per_step"], line["roofline"]["avg_launch_ms"], line["roofline"]["frac"]),
       "# unprofiled bench lines of the same gpurun call:",
       "#   P4: value %.2f MVoxels/s, %.3f ms/step, back avg %.3f ms, roofline.frac %.4f (f32 peak), cpu_baseline %s" % (
           b4["value"], b4["ms_per_step"], b4["roofline"]["avg_launch_ms"], b4["roofline"]["frac"], json.dumps(b4.get("cpu_baseline"))),
       "#   S8: value %.2f MVoxels/s, %.3f ms/step, back avg %.3f ms, roofline.frac %.4f (f32 peak)" % (
           s8["value"], s8["ms_per_step"], s8["roofline"]["avg_launch_ms"], s8["roofline"]["frac"]),
       ]
d = [r[0] / 1e3 for r in c.execute("select (end - start) from kernels where name like '%back_plane_kernel%' order by start").fetchall()]
k = line["steps"]
out += ["# back_plane_kernel dispatches in order, us: " + " ".join("%.0f" % v for v in d),
        "#   the first %d are the Gaussian pass 2, the tile-shape timing (two 4x4 and two slower 4x2 launches) and the warm-up; the %d timed" % (len(d) - k, k),
        "#   steps are the last %d: average %.1f us = the HIP-event figure of the bench line (%.1f us)" % (k, sum(d[-k:]) / k, 1e3 * line["roofline"]["avg_launch_ms"]),
        "%7s %14s %12s %7s  %s" % ("calls", "total_us", "avg_us", "pct", "kernel")]
out += ["%7d %14.1f %12.1f %7.2f  %s" % (k, t, a, p, n) for n, k, t, a, p in rows]
open(f'{R}/profiles/r01_g_final_kernel_stats.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out[:14]))
