import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, bench, sys
from mellon_amd import _lib
ctx=_lib.default_context()
x=bench.gaussian_mixture(1000000,50,3); xd=ctx.to_device(x)
m=int(sys.argv[1]) if len(sys.argv)>1 else 300
ctx.kmeans(xd,m,seed=42,max_iter=0,init="sklearn")
t0=time.perf_counter(); ctx.kmeans(xd,m,seed=42,max_iter=0,init="sklearn"); print("sklearn seeding m=%d:"%m, round(time.perf_counter()-t0,3),"s")
