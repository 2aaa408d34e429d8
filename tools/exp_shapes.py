"""dev tool: the gather and the scatter on P4 for explicit tile shapes (timers of the engine, last of 4 launches).
usage: python tools/exp_shapes.py [workload]"""
import sys; sys.path.insert(0, '/root/repo')
from fetalreconstruction_amd import workloads, engine
from tests.twins.reconstruction import irtkReconstruction
wl = sys.argv[1] if len(sys.argv) > 1 else 'P4'
P = workloads.get(wl)
rec = engine.Reconstruction(0); engine.sync_gpu(rec, P)
d = irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity); d.SetSmoothingParameters(150, 0.02)
d.InitializeEMValuesGPU(); d.GaussianReconstructionGPU(); d.SimulateSlicesGPU(); d.InitializeRobustStatisticsGPU(); d.EStepGPU()
sw = d._local(d._slice_weight_gpu)
rec.timer_enable(True)
for w, h in ((6, 4), (6, 5), (5, 5), (8, 4), (4, 6), (5, 6), (7, 4), (4, 8), (4, 4)):          # at most 32 pixels (FWDU_MAXPIX)
    rec.set_option("fwd_tile_h", 1); rec.set_option("fwd_tile_w", w); rec.set_option("fwd_tile_h", h)
    for _ in range(3):
        rec.timer_reset(); rec.SimulateSlices()
    print(f"gather {w}x{h}: {rec.timers()['forward'][0]:.3f} ms", flush=True)
for w, h in ((4, 4), (5, 4), (6, 4), (6, 5), (8, 4), (7, 4)):          # at most 32 pixels unless built with -DSVR_WAVE_MAXPIX_EVAL=64
    rec.set_option("tile_h", 1); rec.set_option("tile_w", w); rec.set_option("tile_h", h)
    for cap in (2416, 2096):
        rec.set_option("wave_cap", cap)
        for _ in range(3):
            rec.timer_reset(); rec.SuperresolutionBackproject(sw)
        c = rec.counters()
        print(f"scatter {w}x{h} box {cap}: {rec.timers()['backproject'][0]:.3f} ms (rerun8 {c['rerun8_tiles']}, fallback {c['fallback_tiles']})", flush=True)
