"""Round 3: the constants of the two-step reciprocal square root of the canonical PSF sequence.
A Newton-form step y <- y (k - (q/2) y^2) with k = 1.5 + delta centres the error band of the step on zero; the start constant
and both k are searched in float32 arithmetic over q in [1, 4) for the smallest maximum relative error after two steps."""
import numpy as np
f32=np.float32
def run(q, magic, k1, k2, steps):
    y=(np.uint32(magic)-(q.view(np.uint32)>>np.uint32(1))).view(f32)
    h=(f32(0.5)*q).astype(f32)
    for k in steps:
        t=(h*y).astype(f32)
        u=(-(t.astype(np.float64))*y.astype(np.float64)+np.float64(f32(k))).astype(f32)
        y=(y*u).astype(f32)
    return y
q=np.linspace(1,4,200001,dtype=np.float64).astype(f32)
ex=1/np.sqrt(q.astype(np.float64))
def err(magic,k1,k2):
    y=run(q,magic,k1,k2,(k1,k2)).astype(np.float64)
    e=y/ex-1
    return np.abs(e).max(), e.min(), e.max()
# current: 3 NR steps
y=run(q,0x5f375a86,1.5,1.5,(1.5,1.5,1.5)).astype(np.float64); print("3 NR:", np.abs(y/ex-1).max())
y=run(q,0x5f375a86,1.5,1.5,(1.5,1.5)).astype(np.float64); print("2 NR:", np.abs(y/ex-1).max())
best=None
for magic in range(0x5f375a86-0x40000, 0x5f375a86+0x40000, 0x2000):
    # e0 range
    y0=(np.uint32(magic)-(q.view(np.uint32)>>np.uint32(1))).view(f32).astype(np.float64)
    e0=np.abs(y0/ex-1).max()
    d1=0.75*e0*e0
    k1=1.5+d1
    e1=1.5*e0*e0-d1
    d2=0.75*e1*e1
    for s1 in (0.9,1.0,1.1):
      for s2 in (0.8,1.0,1.2):
        r=err(magic,1.5+d1*s1,1.5+d2*s2)
        if best is None or r[0]<best[0]: best=(r[0],magic,1.5+d1*s1,1.5+d2*s2,r)
print(best, hex(best[1]))
best2=None
for magic in range(0x5f377a86-0x3000, 0x5f377a86+0x3000, 0x200):
    for k1 in np.arange(1.5006,1.5012,0.00002):
        for k2i in range(0,12):
            k2=float(f32(1.5)+f32(k2i)*np.spacing(f32(1.5)))
            r=err(magic,float(f32(k1)),k2)
            if best2 is None or r[0]<best2[0]: best2=(r[0],magic,float(f32(k1)),k2)
print(best2, hex(best2[1]), f32(best2[2]).view(np.uint32), f32(best2[3]).view(np.uint32))
