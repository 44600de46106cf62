"""Exact 1-NN on the device: the fp16 pre-filter + fp64 certification against the plain fp64 search.
Sizes: n cells x 50 dims (default 1e6); also a sharded-style call (rows of a slice against all cells) and k-means timing."""
import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from mellon_amd import _lib
ctx = _lib.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = bench.gaussian_mixture(n, 50, 3); xd = ctx.to_device(x)
for rep in range(2):
    t0 = time.perf_counter(); nn = ctx.nn_distances(xd); dt = time.perf_counter() - t0
os.environ["MELLON_AMD_NN_PREFILTER_MIN"] = str(1 << 62)          # read once per process: second process for the plain path
idx = np.random.default_rng(0).choice(n, 40, replace=False)
ref = []
for i in idx:
    d2 = ((x - x[i]) ** 2).sum(1); d2[i] = np.inf; ref.append(np.sqrt(d2.min()))
ref = np.array(ref)
print(f"nn {n} x 50 (prefilter): {dt:.3f} s; max rel err on 40 rows vs brute force {np.abs(nn[idx] - ref).max() / ref.max():.2e}", flush=True)
# slice of rows against all cells (the sharded call), offset self-exclusion
lo, hi = n // 3, n // 3 + 100_000
part = ctx.nn_distances(np.ascontiguousarray(x[lo:hi]), xd, self_offset=lo)
print("slice vs full: max abs diff", float(np.abs(part - nn[lo:hi]).max()), flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "full":
    import subprocess
    out = subprocess.run([sys.executable, "-c",
        "import os,sys,time;sys.path.insert(0,'.');os.environ['MELLON_AMD_NN_PREFILTER']='0';import numpy as np,bench;from mellon_amd import _lib;"
        f"ctx=_lib.default_context();x=bench.gaussian_mixture({n},50,3);xd=ctx.to_device(x);ctx.nn_distances(xd);t0=time.perf_counter();nn=ctx.nn_distances(xd);"
        "print('plain', time.perf_counter()-t0);np.save('/tmp/nn_plain.npy', nn)"], capture_output=True, text=True)
    print(out.stdout.strip(), out.stderr.strip()[-300:])
    plain = np.load("/tmp/nn_plain.npy")
    print("prefilter vs plain fp64 search: max rel diff", float(np.abs(plain - nn).max() / plain.max()), "identical rows", int((plain == nn).sum()), "of", n)
t0 = time.perf_counter(); c, it, inertia = ctx.kmeans(xd, 5000, seed=42, return_info=True); dt = time.perf_counter() - t0
print(f"kmeans {n} x 50, 5000 centres: {dt:.2f} s, {it} sweeps, inertia {inertia:.6g}", flush=True)
