import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from fetalreconstruction_amd import phantom, engine
from fetalreconstruction_amd.reconstruction import irtkReconstruction
P=phantom.problem_p4()
rec=engine.Reconstruction(0); engine.sync_gpu(rec,P)
d=irtkReconstruction(rec,P.ns,max_intensity=P.max_intensity,min_intensity=P.min_intensity); d.SetSmoothingParameters(150,0.02)
d.InitializeEMValuesGPU(); d.GaussianReconstructionGPU(); d.SimulateSlicesGPU(); d.InitializeRobustStatisticsGPU(); d.EStepGPU()
rec.timer_enable(True)
print(rec.counters())
def run(tag):
    rec.SuperresolutionBackproject(d._local(d._slice_weight_gpu))
    rec.timer_reset()
    for _ in range(3): rec.SuperresolutionBackproject(d._local(d._slice_weight_gpu))
    t=rec.timers()['backproject']; c=rec.counters(); print(tag,'ms %.2f'%(t[0]/t[1]), 'tiles',c['tiles'],'fb',c['fallback_tiles'])
rec.set_option("dbg_fwd_lds", 0)
rec.set_option("back_mode",2)
for (tw,th) in ((8,4),(4,4),(4,2),(2,2),(8,2)):
    rec.set_option("tile_w",tw); rec.set_option("tile_h",th)
    for nw,cap in ((8,9600),(6,9600),(8,6272),(6,6272)):
        rec.set_option("plane_waves",nw); rec.set_option("plane_cap",cap)
        run(f"tile {tw}x{th} waves {nw} cap {cap}")
