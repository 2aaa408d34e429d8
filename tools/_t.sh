python -m pytest tests/test_pvr.py -m gpu -x -q -k "launch_order or cell_sizes" 2>&1 | grep -E "passed|failed|rror" | tail -5
for w in P4 S8 PVR8spx; do
for c in 0 1; do
python - <<EOF
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from tools.shard_probe import build, make_engine
from fetalreconstruction_amd import engine as E
wl="$w"; P=build(wl); pvr=wl.startswith("PVR")
rec=make_engine(P, pvr, getattr(P,"spx_masks",None), [("cell_combine",$c)])
ones=np.ones(P.ns,np.float32); rec.UpdateScaleVector(ones,ones); rec.InitializeEMValues(); rec.GaussianReconstruction(); rec.SimulateSlices()
rng=np.random.default_rng(0)
rec.debug_set(E.BUF_WEIGHTS, np.where(P.slices != -1, rng.uniform(0.2, 1.0, P.slices.shape), 0).astype(np.float32))
rec.SuperresolutionBackproject(ones); rec.timer_enable(True); rec.timer_reset()
for _ in range(5): rec.SuperresolutionBackproject(ones)
t=rec.timers()["backproject"]; a=rec.debug_get(E.BUF_ADDON); c=rec.debug_get(E.BUF_CONFIDENCE_MAP)
import hashlib
print(wl, "cell_combine", $c, "scatter %.3f ms"%(t[0]/t[1]), hashlib.md5(a.tobytes()+c.tobytes()).hexdigest()[:12], flush=True)
EOF
done; done
