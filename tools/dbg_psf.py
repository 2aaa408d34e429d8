import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from fetalreconstruction_amd import phantom, engine
from oracle import pyoracle as po
P=phantom.problem_tiny()
rec=engine.Reconstruction(0); engine.sync_gpu(rec,P); rec.UpdateScaleVector(np.ones(P.ns),np.ones(P.ns))
o=po.OracleReconstruction(P,po.CANON)
act=np.argwhere(P.slices!=-1)
rng=np.random.default_rng(0)
nbad=0
for i in rng.choice(len(act),300,replace=False):
    sl,py,px=act[i]
    v,c=rec.probe_pixel(sl,px,py)
    n,bits,vals,cc=o.tap_census(sl,px,py,with_vals=True)
    raw=o.psf_values(sl,px,py)
    if not np.array_equal(c,cc.astype(np.int32)): print('centre diff',c,cc)
    kept_d = ~(v<0)
    kept_o = ~(vals<0)
    d=np.where(kept_d&kept_o, np.abs(v-vals),0)
    if (kept_d!=kept_o).any() or d.max()>0:
        nbad+=1
        j=np.argmax(np.abs(np.where(kept_d,v,raw)-raw))
        if nbad<6:
            print('pixel',sl,px,py,'flips',(kept_d!=kept_o).sum(),'max val diff',d.max(), 'at',j, v[j], raw[j], 'n mism vals', (np.where(kept_d,v,raw)!=raw).sum())
print('bad pixels',nbad,'of 300')
