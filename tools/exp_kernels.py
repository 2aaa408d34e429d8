"""dev tool: min-of-N timings of the hot kernels on P4 for the library named by SVR_HIP_LIB.
usage: [SVR_HIP_LIB=...] python tools/exp_kernels.py [opt=value ...]"""
import os, sys; sys.path.insert(0, '/root/repo')
from fetalreconstruction_amd import phantom, engine
from fetalreconstruction_amd.reconstruction import irtkReconstruction
P = phantom.problem_p4()
if os.environ.get('STACK'):
    import numpy as np
    P = phantom.sub_problem(P, 0, 0, select=np.where(P.stack_index == int(os.environ['STACK']))[0])
rec = engine.Reconstruction(0); engine.sync_gpu(rec, P)
for a in sys.argv[1:]:
    k, v = a.split('='); rec.set_option(k, int(v))
d = irtkReconstruction(rec, P.ns, max_intensity=P.max_intensity, min_intensity=P.min_intensity); d.SetSmoothingParameters(150, 0.02)
d.InitializeEMValuesGPU(); d.GaussianReconstructionGPU(); d.SimulateSlicesGPU(); d.InitializeRobustStatisticsGPU(); d.EStepGPU()
rec.timer_enable(True)
def best(f, name, n=6):
    f(); ts = []
    for _ in range(n):
        rec.timer_reset(); f(); t = rec.timers()[name]; ts.append(t[0] / t[1])
    return min(ts)
sw = d._local(d._slice_weight_gpu)
res = dict(fwd=best(rec.SimulateSlices, 'forward'), back=best(lambda: rec.SuperresolutionBackproject(sw), 'backproject'),
           gauss=best(rec.GaussianReconstruction, 'gauss', 3))
c = rec.counters()
print('tiles', c['tiles'], 'fallback', c['fallback_tiles'])
print(os.path.basename(os.environ.get('SVR_HIP_LIB', 'libsvr_hip.so')), ' '.join(sys.argv[1:]), ' '.join(f'{k} {v:.2f}' for k, v in res.items()))
