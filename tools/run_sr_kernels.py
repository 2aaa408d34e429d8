"""dev tool: a few launches of the scatter and the gather on a workload with the given options (for rocprofv3 passes).
usage: python tools/run_sr_kernels.py [workload] [opt=value ...]"""
import sys; sys.path.insert(0, '/root/repo')
from fetalreconstruction_amd import workloads, engine
from fetalreconstruction_amd.reconstruction import irtkReconstruction
args = sys.argv[1:]
wl = args.pop(0) if args and '=' not in args[0] else 'P4'
P = workloads.get(wl)
rec = engine.Reconstruction(0); engine.sync_gpu(rec, P)
for a in args:
    k, v = a.split('='); rec.set_option(k, int(v))
d = irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity); d.SetSmoothingParameters(150, 0.02)
d.InitializeEMValuesGPU(); d.GaussianReconstructionGPU(); d.SimulateSlicesGPU(); d.InitializeRobustStatisticsGPU(); d.EStepGPU()
sw = d._local(d._slice_weight_gpu)
for _ in range(40):          # many more than the tuners' trial launches: pmc.py averages the last half of a kernel's dispatches
    rec.SuperresolutionBackproject(sw)
for _ in range(40):
    rec.SimulateSlices()
print("tuned: scatter tiles %dx%d box %d, gather tiles %dx%d box %d" % tuple(rec.get_option(k) for k in ("tile_w", "tile_h", "wave_cap", "fwd_tile_w", "fwd_tile_h", "fwd_unit_cap")), flush=True)
