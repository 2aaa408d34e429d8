import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from fetalreconstruction_amd import phantom, engine
from fetalreconstruction_amd.reconstruction import irtkReconstruction
P=phantom.problem_p4()
rec=engine.Reconstruction(0); engine.sync_gpu(rec,P)
d=irtkReconstruction(rec,P.ns,max_intensity=P.max_intensity,min_intensity=P.min_intensity); d.SetSmoothingParameters(150,0.02)
d.InitializeEMValuesGPU(); d.GaussianReconstructionGPU(); d.SimulateSlicesGPU(); d.InitializeRobustStatisticsGPU(); d.EStepGPU()
for _ in range(2):
    rec.SuperresolutionBackproject(d._local(d._slice_weight_gpu)); rec.SimulateSlices()
