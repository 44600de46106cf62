#!/usr/bin/env python
"""bench.py -- cells/sec of DensityEstimator.fit_predict on MI355X (BASELINE.json metric).

One "step" = one complete fit_predict pass of the hot path over the synthetic workload
(covariance tiles -> Cholesky -> triangular-solve panels -> Ridge init -> L-BFGS-B MAP solve on
the fused device objective -> log-density), inputs already resident in HBM when the timed
region starts.  Workload: BASELINE config 3 -- 1e6 cells x 50 dims Gaussian mixture, 5 000 landmarks,
Matern52 -- per GPU: ONE model is fitted on all N x 1e6 cells, cell-sharded over the N ranks (weak
scaling: fixed cells per GPU; one process per GPU, RCCL all-reduce of (loss, grad) per evaluation and
of the Ridge Gram once per fit).  `--scaling strong` shards the same 1e6 cells over the N ranks instead
(Amdahl-limited by the replicated m x m factorisations).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (fields documented in DESIGN.md S6).
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 achievable


def gaussian_mixture(n, d, seed, k=10, shard=0):
    """BASELINE.md S2 synthetic cells: 10 isotropic Gaussian components, PCG64(seed), float64.
    shard > 0: n further cells of the SAME mixture from an independent stream (weak scaling)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    means = rng.normal(0.0, 3.0, size=(k, d))
    sig = rng.uniform(0.5, 1.5, size=k)
    if shard > 0:
        rng = np.random.Generator(np.random.PCG64([seed, shard]))
    comp = rng.integers(0, k, size=n)
    x = means[comp] + rng.normal(size=(n, d)) * sig[comp][:, None]
    return np.ascontiguousarray(x[rng.permutation(n)])


def make_landmarks(x, m, seed=42, sub=20000, iters=10):
    """k-means centroids on a subsample (the reference recommends subset k-means at this scale,
    base_model.py:227-233).  Shared input, computed once on the host, excluded from timing."""
    from sklearn.cluster import k_means
    rng = np.random.default_rng(seed)
    idx = rng.choice(x.shape[0], size=min(sub, x.shape[0]), replace=False)
    if m >= idx.size:
        return np.ascontiguousarray(x[idx[:m]])
    c = k_means(x[idx], m, n_init=1, random_state=seed, max_iter=iters, init="random")[0]
    # centroids differ in their last bits from process to process (BLAS / OpenMP code paths), enough to move the
    # L-BFGS pass count by a few evaluations; rounded through float32 they are the same numbers in every run
    return np.ascontiguousarray(c.astype(np.float32).astype(np.float64))


def cpu_baseline(x, landmarks, nn, kern_name, sample):
    """The oracle (NumPy/SciPy restatement of the reference's JAX-CPU path, reference stopping
    rule) timed on this box's host cores on the first `sample` cells of the SAME workload."""
    from oracle import mellon_oracle as mo
    xs, nns = x[:sample], nn[:sample]
    t0 = time.perf_counter()
    fit = mo.density_fit(xs, cov_func_curry=getattr(mo, kern_name), landmarks=landmarks, nn_distances=nns)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    return {"value": sample / dt, "unit": "cells/s", "cores": int(threads), "kind": "port",
            "sample": f"first {sample} cells of the workload, m={landmarks.shape[0]}, d={xs.shape[1]}, "
                      f"{fit.n_eval} objective evaluations (reference L-BFGS-B defaults), {dt:.1f} s wall, "
                      f"os.cpu_count()={os.cpu_count()}"}, fit


def _stdout_to_stderr():
    """Route file descriptor 1 to stderr and return a handle on the real stdout.  Libraries loaded by the run print
    to the C-level stdout on their own ("[Gloo] Rank 0 is connected ...", RCCL's version banner at communicator
    set-up, which C stdio only flushes at exit, i.e. AFTER Python's output); the contract is ONE JSON line there."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def _print_result_line(real_stdout, line):
    """Flush everything buffered so far to stderr, emit `line` alone on the real stdout, then keep stdout pointed at
    stderr for whatever libraries print while shutting down."""
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    os.write(real_stdout, (line + "\n").encode())
    os.close(real_stdout)


def main():
    real_stdout = _stdout_to_stderr()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", dest="n", type=int, default=1_000_000, help="cells per GPU (weak) / in total (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--dims", dest="d", type=int, default=50)
    ap.add_argument("--landmarks", dest="m", type=int, default=5000)
    ap.add_argument("--kernel", default="Matern52")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=50000, help="cells of the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
        args.gpus = world

    import mellon_amd
    from mellon_amd import _lib, distributed
    comm = distributed.init_from_env()
    ctx = _lib.default_context()
    info = ctx.device_info()

    # ---- synthetic workload -----------------------------------------------------------------------------
    n, d, m = args.n, args.d, args.m
    weak = args.scaling == "weak"
    t_gen = time.perf_counter()
    x0 = gaussian_mixture(n, d, args.seed)                # shard 0 == the BASELINE C3 data set
    # single-threaded k-means (3.6 s, untimed): bit-reproducible landmarks -- with threads they differ in the last
    # bits from run to run, which is enough to move the L-BFGS pass count by a few evaluations -- and no
    # oversubscription of the host when N ranks share it
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):
        landmarks = make_landmarks(x0, m)
    if world > 1:   # replicated inputs must be BIT-identical on every rank (they steer the shared optimiser)
        landmarks = comm.allreduce_sum(landmarks if rank == 0 else np.zeros_like(landmarks))
    t_nn = 0.0
    if weak:
        # rank r owns shard r (n cells); exact 1-NN among ALL world * n cells: every rank regenerates the
        # other shards (deterministic streams, no communication) and keeps the running minimum
        n_total = n * world
        lo, hi = 0, n
        x_loc = x0 if rank == 0 else gaussian_mixture(n, d, args.seed, shard=rank)
        x_loc_dev = ctx.to_device(x_loc)
        nn_loc = None
        for s_ in range(world):
            xs = x_loc if s_ == rank else (x0 if s_ == 0 else gaussian_mixture(n, d, args.seed, shard=s_))
            xs_dev = x_loc_dev if s_ == rank else ctx.to_device(xs)
            t0 = time.perf_counter()
            part = ctx.nn_distances(x_loc_dev, xs_dev, self_offset=0 if s_ == rank else -(n + 1))
            t_nn += time.perf_counter() - t0
            nn_loc = part if nn_loc is None else np.minimum(nn_loc, part)
            if s_ != rank:
                xs_dev.free()
            del xs
    else:
        n_total = n
        lo, hi = distributed.shard_bounds(n, world, rank)
        x_loc = x0[lo:hi]
        x_all_dev = ctx.to_device(x0)
        x_loc_dev = ctx.to_device(x_loc) if world > 1 else x_all_dev
        t0 = time.perf_counter()
        nn_loc = ctx.nn_distances(x_loc_dev, x_all_dev, self_offset=lo)      # exact 1-NN, excluded from timing
        t_nn = time.perf_counter() - t0
        if world > 1:
            x_all_dev.free()
    t_gen = time.perf_counter() - t_gen
    kern = getattr(mellon_amd.cov, args.kernel)

    def one_step():
        est = mellon_amd.DensityEstimator(cov_func_curry=kern, landmarks=landmarks, nn_distances=nn_loc)
        dens = est.fit_predict(x_loc_dev)
        return est, dens

    def fence():
        comm.barrier()
        ctx.synchronize()

    def release(est):
        """Return the fit's device buffers (to the library's cache) deterministically."""
        est._fit.close()

    for _ in range(args.warmup):
        est, dens = one_step()
        release(est)
        del est
    gc.collect()
    fence()
    t_fit = t_free = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ta = time.perf_counter()
        est, dens = one_step()
        tb = time.perf_counter()
        stats = est._fit.stage_times()
        n_eval = est.loss_func.n_eval
        if _ + 1 < args.steps:
            release(est)
            del est
        t_fit += tb - ta
        t_free += time.perf_counter() - tb
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = float(comm.allreduce_sum(np.eye(world)[rank] * elapsed).max())   # MAX over ranks

    # ---- size-independent parity property at full size: predict(X) == fit_predict(X) -----------------
    k = min(20000, hi - lo)
    xq = x_loc[:k]
    prop = float(np.abs(est.predict(xq) - dens[:k]).max() / np.abs(dens[:k]).max())

    if rank != 0:
        return
    ms_per_step = 1e3 * elapsed / args.steps
    value = n_total * args.steps / elapsed
    per_launch = stats["objective_kernel_s"] / max(stats["objective_launches"], 1.0)
    ach = stats["objective_bytes_per_launch"] / per_launch / 1e9
    # fp32 warm-up passes of the MAP solve (mixed precision): same rows, 4 bytes per element
    n32 = stats.get("objective32_launches", 0.0)
    per_launch32 = stats.get("objective32_kernel_s", 0.0) / max(n32, 1.0)
    bytes32 = stats["objective_bytes_per_launch"] / 2.0
    ach32 = bytes32 / per_launch32 / 1e9 if n32 > 0 else 0.0
    traffic = traffic32 = None
    tfile = os.path.join(ROOT, "profiles", "objective_traffic.json")
    if os.path.exists(tfile):
        try:
            t = json.load(open(tfile))
            if t.get("n_local") == hi - lo and t.get("m") == m:
                traffic = t.get("hbm_bytes_per_launch")
                traffic32 = t.get("fp32_passes", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = traffic32 = None
    out = {
        "metric": "cells/sec fit_predict", "value": value, "unit": "cells/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C3 DensityEstimator.fit_predict: {n_total} cells x {d} dims Gaussian mixture "
                               f"(seed {args.seed}), {m} landmarks, {args.kernel}, one model, cells sharded over "
                               f"{world} GPU(s) ({hi - lo} cells per GPU)",
                   "n": n_total, "n_per_gpu": hi - lo, "d": d, "m": m, "kernel": args.kernel,
                   "parallelism": f"cells/{world}",
                   "objective_evaluations": int(n_eval), "objective_evaluations_fp32": int(n32),
                   "optimizer": "L-BFGS-B maxcor=30 ftol=1e-13 gtol=1e-7 (converged to the unique MAP optimum)",
                   "landmarks": "k-means (random init, 10 Lloyd iterations, 20k-cell subsample), host, untimed",
                   "nn_distances": f"exact 1-NN on device, untimed ({t_nn:.2f} s)",
                   "predict_equals_fit_predict_rel_max": prop, "device": info["arch"]},
        "roofline": {"bound": "hbm", "kernel": "k_objective (fused loss+grad, one pass over L)",
                     "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": traffic, "algorithmic_bytes_per_launch": stats["objective_bytes_per_launch"],
                     "avg_launch_ms": 1e3 * per_launch, "launches": int(stats["objective_launches"]),
                     "share_of_step": stats["objective_kernel_s"] / (elapsed / args.steps) if world == 1 else None},
        "roofline_fp32_passes": {"bound": "hbm", "kernel": "k_objective32 (same pass over the fp32 copy of K: warm-up "
                                 "iterations of the MAP solve)", "achieved": ach32, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": ach32 / HBM_PEAK_GBS, "traffic": traffic32,
                                 "algorithmic_bytes_per_launch": bytes32, "avg_launch_ms": 1e3 * per_launch32,
                                 "launches": int(n32)},
        "stages_s": {k: round(v, 4) for k, v in stats.items() if k.endswith("_s")},
        "host_s": {"fit_predict_per_step": round(t_fit / args.steps, 4), "release_per_step": round(t_free / args.steps, 4)},
    }
    if world == 1 and args.cpu_sample > 0:
        del est
        gc.collect()
        base, _ = cpu_baseline(x0, landmarks, nn_loc, args.kernel, min(args.cpu_sample, n))
        out["cpu_baseline"] = base
    _print_result_line(real_stdout, json.dumps(out))


if __name__ == "__main__":
    main()
