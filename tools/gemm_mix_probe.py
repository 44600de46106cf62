"""k_dgemm probe: the shapes of a C3 preconditioner build and of C5, under the tile policies of launch_dgemm.
Run once per policy (the switches are read once per process):
  MELLON_AMD_EXPERIMENTAL=1 MELLON_AMD_GEMM_MIX=0 python tools/gemm_mix_probe.py     # single-size kernels (round 3)
  MELLON_AMD_EXPERIMENTAL=1 MELLON_AMD_GEMM_MIX=1 python tools/gemm_mix_probe.py     # pipelined, whole rounds 128 + quadrants
  ... =2 all 128 pipelined, =3 all quadrants pipelined"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def child():
    from mellon_amd import _lib
    ctx = _lib.default_context()
    kmode = int(os.environ.get("MELLON_AMD_DIAG_KMODE", "0"))
    m = 5000
    shapes = [
        # (label, ta, tb, M, N, K, lower_only, flops)
        ("NT  5000^3 full", 0, 1, m, m, m, 0, 2.0 * m ** 3),
        ("NN  5000^3 full", 0, 0, m, m, m, 0, 2.0 * m ** 3),
        ("NT  8192^3 full", 0, 1, 8192, 8192, 8192, 0, 2.0 * 8192 ** 3),
        ("NT  syrk lower 4744^2 K=256", 0, 1, 4744, 4744, 256, 1, 4744.0 ** 2 * 256),
        ("NT  syrk lower 2560^2 K=256", 0, 1, 2560, 2560, 256, 1, 2560.0 ** 2 * 256),
        ("NT  panel 2e5 x 2000 K=2000", 0, 1, 200000, 2000, 2000, 0, 2.0 * 2e5 * 2000 * 2000),
        ("NN  predict 2e5 x 2000 K=2000", 0, 0, 200000, 2000, 2000, 0, 2.0 * 2e5 * 2000 * 2000),
        ("NT  3000^3", 0, 1, 3000, 3000, 3000, 0, 2.0 * 3000 ** 3),
        ("NT  2000^3", 0, 1, 2000, 2000, 2000, 0, 2.0 * 2000 ** 3),
    ]
    if kmode:
        shapes = [("NT 5000^3 kmode %d" % kmode, 0, 1, m, m, m, 0, (1.0 if kmode in (3, 4) else 1.0 / 3) * m ** 3),
                  ("NN 5000^3 kmode %d" % kmode, 0, 0, m, m, m, 0, (1.0 if kmode in (3, 4) else 1.0 / 3) * m ** 3)]
    for label, ta, tb, M, N, K, lo, fl in shapes:
        ms = ctx.diag_dgemm(ta, tb, M, N, K, lower_only=lo, reps=5)
        print(f"  {label:34s} {ms:8.3f} ms  {fl / ms / 1e9:6.1f} TF/s  ({fl / ms / 1e9 / 78.6:.2f})", flush=True)


def correctness():
    from mellon_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(0)
    worst = 0.0
    for (M, N, K, ta, tb) in [(3000, 3100, 700, False, True), (2945, 3333, 515, True, False), (4100, 2900, 129, False, False),
                              (3072, 3072, 48, True, True), (5000, 5000, 33, False, True)]:
        A = rng.normal(size=(K, M) if ta else (M, K))
        B = rng.normal(size=(N, K) if tb else (K, N))
        want = (A.T if ta else A) @ (B.T if tb else B)
        got = ctx.gemm(A, B, ta=ta, tb=tb).to_host()
        err = np.abs(got - want).max() / np.abs(want).max()
        acc = ctx.to_device(np.ones((M, N)))
        ctx.gemm(ctx.to_device(A), B, ta=ta, tb=tb, alpha=0.5, beta=2.0, out=acc)
        err2 = np.abs(acc.to_host() - (0.5 * want + 2.0)).max() / np.abs(want).max()
        worst = max(worst, err, err2)
        print(f"  gemm {M}x{N}x{K} ta={ta} tb={tb}: rel err {err:.2e} / {err2:.2e}", flush=True)
    # factorisation + solves at a size where the updates run through the mixed kernel
    m = 5000
    X = rng.normal(size=(m, 40))
    S = X @ X.T + m * np.eye(m)
    L = ctx.chol_lower(S)
    if L is not None:
        err = np.abs(L @ L.T - S).max() / np.abs(S).max()
        worst = max(worst, err)
        print(f"  chol({m}): |L L^T - S| rel {err:.2e}", flush=True)
    print("  worst", worst, "OK" if worst < 1e-12 else "FAIL", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    elif len(sys.argv) > 1 and sys.argv[1] == "check":
        correctness()
    else:
        for mix in os.environ.get("PROBE_MODES", "0 1 2 3").split():
            for km in os.environ.get("PROBE_KMODES", "0 3 7").split():
                env = dict(os.environ, MELLON_AMD_EXPERIMENTAL="1", MELLON_AMD_GEMM_MIX=mix, MELLON_AMD_DIAG_KMODE=km)
                print(f"== MELLON_AMD_GEMM_MIX={mix} kmode={km}", flush=True)
                subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
        env = dict(os.environ, MELLON_AMD_EXPERIMENTAL="1", MELLON_AMD_GEMM_MIX="1")
        print("== correctness (mix = 1)", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "check"], env=env)
