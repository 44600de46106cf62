"""Remez minimax polynomials of e^t on [-ln2/2, ln2/2] (relative error), degrees 9..11, in long double: the coefficients of
covepi::exp_neg (mellon_amd/csrc/cov_epilogue.h) are the degree-11 row.  python tools/exp_poly_remez.py"""
import numpy as np
from numpy.polynomial import chebyshev as C
LD=np.longdouble
a=LD(np.log(LD(2)))/2
def f(t): return np.exp(t)
def remez(n, iters=30):
    # nodes: chebyshev extrema
    k=np.arange(n+2)
    x=(a*np.cos(np.pi*k/(n+1))).astype(LD)[::-1]
    for it in range(iters):
        # solve sum c_j x^j + (-1)^i E * w(x_i) = f(x_i), w = f (relative error)
        A=np.zeros((n+2,n+2),dtype=LD)
        for j in range(n+1): A[:,j]=x**j
        A[:,n+1]=((-1)**k)*f(x)
        sol=np.linalg.solve(A.astype(np.float64),f(x).astype(np.float64)).astype(LD)
        # refine in long double via iterative refinement
        for _ in range(5):
            r=f(x)-A@sol
            sol=sol+np.linalg.solve(A.astype(np.float64),r.astype(np.float64)).astype(LD)
        c=sol[:n+1]; E=sol[n+1]
        # find extrema of relative error on fine grid
        g=np.linspace(-a,a,200001).astype(LD)
        p=np.zeros_like(g)
        for j in range(n,-1,-1): p=p*g+c[j]
        err=(p-f(g))/f(g)
        # locate n+2 alternating extrema
        idx=[0]
        s=np.sign(err)
        # split by sign changes
        ch=np.nonzero(s[1:]!=s[:-1])[0]+1
        segs=np.split(np.arange(len(g)),ch)
        ext=[seg[np.argmax(np.abs(err[seg]))] for seg in segs]
        if len(ext)!=n+2:
            break
        xn=g[ext]
        if np.max(np.abs(xn-x))<1e-12: x=xn; break
        x=xn
    return c,float(np.max(np.abs(err))),float(E)
for n in (9,10,11):
    c,e,E=remez(n)
    print(n,e,E)
    print([float(v).hex() for v in c])
    print([repr(float(v)) for v in c])
    # check fp64 Horner error incl. rounding on random t
    t=np.random.default_rng(0).uniform(-float(a),float(a),200000)
    cf=[float(v) for v in c]
    p=np.full_like(t,cf[n])
    for j in range(n-1,-1,-1): p=p*t+cf[j]   # not fma but close
    ref=np.exp(t.astype(LD))
    print(' fp64 horner max rel err',float(np.max(np.abs((p.astype(LD)-ref)/ref))))
