import time, sys
sys.path.insert(0, '.')
t0=time.time()
import numpy as np, torch
t1=time.time(); print("import torch %.2f"%(t1-t0))
from fetalreconstruction_amd import engine as E, phantom
tiny = phantom.problem_tiny()
for rep in range(3):
    t=time.time(); rec = E.Reconstruction(0); a=time.time()-t
    t=time.time(); E.sync_gpu(rec, tiny); b=time.time()-t
    t=time.time(); rec.UpdateScaleVector(np.ones(tiny.ns, np.float32), np.ones(tiny.ns, np.float32)); rec.GaussianReconstruction(); c=time.time()-t
    t=time.time(); rec.SimulateSlices(); v=rec.syncCPU(); d=time.time()-t
    t=time.time(); rec.close(); e=time.time()-t
    print("rep %d create %.3f sync_gpu %.3f gauss %.3f sim+sync %.3f close %.3f"%(rep,a,b,c,d,e))
