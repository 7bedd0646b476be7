import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
# fit f(s)=atan(sqrt(s))/sqrt(s) on s in [0,1]
def f(s):
    a=np.sqrt(s); out=np.ones_like(s); m=a>0; out[m]=np.arctan(a[m])/a[m]; return out
for K in (7,8,9,10):
    n=4000
    x=np.cos(np.pi*(np.arange(n)+0.5)/n)  # cheb nodes in [-1,1]
    s=(x+1)/2
    # minimize relative-ish error: weight 1
    ch=C.chebfit(x,f(s),K)
    poly=C.cheb2poly(ch)  # in x
    # convert x=2s-1 to s poly
    ps=np.zeros(1)
    px=P.Polynomial(poly)
    comp=px(P.Polynomial([-1,2]))
    c=comp.coef
    c32=c.astype(np.float32)
    a=np.linspace(0,1,2000001).astype(np.float32)
    s32=(a*a).astype(np.float32)
    r=np.full_like(s32,c32[-1])
    for k in range(len(c32)-2,-1,-1):
        r=(r.astype(np.float64)*s32.astype(np.float64)+c32[k].astype(np.float64)).astype(np.float32)  # fma emu
    r=(r*a).astype(np.float32)
    err=np.abs(r.astype(np.float64)-np.arctan(a.astype(np.float64)))
    print(K,err.max(), [repr(float(v)) for v in c32])
