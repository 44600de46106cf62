import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mellon_amd import _lib
ctx = _lib.default_context()
print("fp64 MFMA peak TF/s:", round(ctx.diag_peak(0), 2), " HBM read GB/s:", round(ctx.diag_peak(1, 16 << 30), 1))
n, m = 1_000_000, 5000
for K in (1280, 2560, 5000):
    ms = ctx.diag_dgemm(0, 1, n, 128, K, reps=3)
    print(f"NT trsm-step  M={n} N=128 K={K}: {ms:.2f} ms  {2*n*128*K/ms/1e9:.1f} TF/s")
ms = ctx.diag_dgemm(0, 1, 8192, 8192, 8192, reps=2)
print(f"NT square 8192^3: {ms:.2f} ms {2*8192**3/ms/1e9:.1f} TF/s")
ms = ctx.diag_dgemm(1, 0, m, m, n, lower_only=1, split_k=16, reps=1)
print(f"TN gram lower M=N={m} K={n} split16: {ms:.2f} ms {n*m*m/ms/1e9:.1f} TF/s (useful)")
ms = ctx.diag_dgemm(1, 0, m, m, n // 8, lower_only=1, split_k=2, reps=1)
print(f"TN gram lower K={n//8} split2: {ms:.2f} ms {n//8*m*m/ms/1e9:.1f} TF/s (useful)")
ms = ctx.diag_dgemm(0, 0, 200000, 2000, 2000, reps=2)
print(f"NN predict 2e5x2000x2000: {ms:.2f} ms {2*2e5*2000*2000/ms/1e9:.1f} TF/s")
