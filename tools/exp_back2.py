"""dev tool: scatter / gather timings on a workload for a list of option sets.
usage: python tools/exp_back2.py [workload] -- "opt=v opt=v" "opt=v" ..."""
import os, sys; sys.path.insert(0, '/root/repo')
import numpy as np
from fetalreconstruction_amd import workloads, engine
from tests.twins.reconstruction import irtkReconstruction
wl = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != '--' else 'P4'
sets = sys.argv[sys.argv.index('--') + 1:] if '--' in sys.argv else ['']
P = workloads.get(wl)
if os.environ.get('STACK'):
    from fetalreconstruction_amd import phantom
    P = phantom.sub_problem(P, 0, 0, select=np.where(P.stack_index == int(os.environ['STACK']))[0])
for s in sets:
    rec = engine.Reconstruction(0); engine.sync_gpu(rec, P)
    for a in s.split():
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
    res = dict(back=best(lambda: rec.SuperresolutionBackproject(sw), 'backproject'), fwd=best(rec.SimulateSlices, 'forward'),
               gauss=best(rec.GaussianReconstruction, 'gauss', 3))
    c = rec.counters()
    print(f"{wl} [{s}] Va {c['Va']} tiles {c['tiles']} fb8 {c.get('rerun8_tiles')} fb {c['fallback_tiles']} | " + ' '.join(f'{k} {v:.3f}' for k, v in res.items()), flush=True)
    rec.close()
