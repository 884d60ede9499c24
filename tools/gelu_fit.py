#!/usr/bin/env python3
"""Fit of the transcendental-free GELU used by the FC1 epilogue (common.h gelu_erf_fast*): minimise the max |gelu error| of
   max(x,0) - a*R(a)^8, a = min(|x|, X0), R a degree-5 polynomial; then check it in emulated fp32 arithmetic."""
import numpy as np
from scipy.special import erfc
from scipy.optimize import least_squares
def gelu_exact(x): return 0.5*x*erfc(-x/np.sqrt(2))
X0=5.5; k=3; deg=5
xs=np.linspace(0,X0,40001)
def err(c):
    rr=np.polyval(c[::-1],xs)**(2**k)      # = 0.5*erfc
    return np.concatenate([-xs*rr-gelu_exact(-xs), xs-xs*rr-gelu_exact(xs)])
tgt=(0.5*erfc(xs/np.sqrt(2)))**(1.0/2**k)
c0=np.polyfit(xs,tgt,deg)[::-1]
for p in (2,4,8,16,32,64):
    res=least_squares(lambda c:(np.abs(err(c))*1e5)**(p/2),c0,method='lm',max_nfev=20000,xtol=1e-15,ftol=1e-15)
    if np.all(np.isfinite(res.x)): c0=res.x
    print(p,np.max(np.abs(err(c0))))
c32=c0.astype(np.float32)
print('coef',[f"{v:.9e}" for v in c32])
# float32 emulation
x=np.linspace(-12,12,2000001).astype(np.float32)
ax=np.minimum(np.abs(x),np.float32(X0))
p=np.full_like(x,c32[5])
for i in (4,3,2,1,0): p=(p*ax+c32[i]).astype(np.float32)
p=(p*p).astype(np.float32);p=(p*p).astype(np.float32);p=(p*p).astype(np.float32)
g=(np.maximum(x,np.float32(0))-ax*p).astype(np.float32)
ex=gelu_exact(x.astype(np.float64))
e=np.abs(g-ex)
print('f32 max abs err',e.max(),'at',x[e.argmax()])
# error relative to fp16 half-ulp of exact value
h=np.abs(ex).astype(np.float16); ulp=np.spacing(h).astype(np.float64)
print('max err / fp16 ulp',np.max(e/np.maximum(ulp,6e-8)))
print('min p', p.min(), 'p at X0', p[-1])
