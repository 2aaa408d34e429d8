"""dev tool: time the forward / back-projection kernels on P4 for tile configurations given as
tw,th,waves,cap arguments, e.g.  python tools/exp_back.py 4,4,8,9600 8,4,8,9600"""
import sys; sys.path.insert(0, '/root/repo')
from fetalreconstruction_amd import phantom, engine
from fetalreconstruction_amd.reconstruction import irtkReconstruction
P = phantom.problem_p4()
rec = engine.Reconstruction(0); engine.sync_gpu(rec, P)
d = irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity); d.SetSmoothingParameters(150, 0.02)
d.InitializeEMValuesGPU(); d.GaussianReconstructionGPU(); d.SimulateSlicesGPU(); d.InitializeRobustStatisticsGPU(); d.EStepGPU()
rec.timer_enable(True)
for mode, tw, th, cap in ((0, 8, 4, 9216), (1, 8, 4, 9216), (1, 8, 8, 12288), (1, 4, 4, 7168), (1, 8, 4, 12288)):
    rec.set_option("fwd_mode", mode); rec.set_option("fwd_tile_w", tw); rec.set_option("fwd_tile_h", th); rec.set_option("fwd_cap", cap)
    rec.SimulateSlices(); rec.timer_reset()
    for _ in range(3): rec.SimulateSlices()
    t = rec.timers()['forward']; print(f'fwd mode {mode} tile {tw}x{th} cap {cap} ms %.2f' % (t[0] / t[1]))
rec.GaussianReconstruction(); rec.timer_reset(); rec.GaussianReconstruction()
t = rec.timers()['gauss']; print('gauss ms %.2f' % (t[0] / t[1]))
rec.set_option("back_mode", 2)
for a in sys.argv[1:]:
    tw, th, nw, cap = [int(v) for v in a.split(',')]
    rec.set_option("tile_w", tw); rec.set_option("tile_h", th); rec.set_option("plane_waves", nw); rec.set_option("plane_cap", cap)
    rec.SuperresolutionBackproject(d._local(d._slice_weight_gpu)); rec.timer_reset()
    for _ in range(3): rec.SuperresolutionBackproject(d._local(d._slice_weight_gpu))
    t = rec.timers()['backproject']; c = rec.counters()
    print(f"tile {tw}x{th} waves {nw} cap {cap} ms %.2f" % (t[0] / t[1]), 'tiles', c['tiles'], 'fb', c['fallback_tiles'])
