#!/usr/bin/env python
"""bench.py -- cells/sec of DensityEstimator.fit_predict on MI355X (BASELINE.json metric).

One "step" = one complete fit_predict pass of the hot path over the synthetic workload
(covariance tiles -> Cholesky -> Ridge init / preconditioner -> L-BFGS MAP solve on the fused device
objective -> log-density), inputs already resident in HBM when the timed region starts.
Workload: BASELINE config 3 -- 1e6 cells x 50 dims Gaussian mixture, 5 000 landmarks, Matern52.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N > 1: ONE model on the SAME 1e6 cells, cell-sharded over the N ranks (`"scaling": "strong"`, the BASELINE config;
one process per GPU, RCCL all-reduce of (loss, grad) per evaluation and of the Ridge Gram once per fit).
`--scaling weak` fits one model on N x 1e6 cells instead (1e6 per GPU).  Only the launcher's environment variables
(RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) are used; the ranks' host sides talk over a Unix socket.

Prints ONE JSON line on rank 0 (fields documented in DESIGN.md S6).  Besides the K timed steps the run measures,
untimed by the contract but in the same process, the pure-fp64 step (`ms_per_step_fp64_only`) and the
host-to-host step (`ms_per_step_host_to_host`, x uploaded inside the step).
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
FP64_MFMA_PEAK_TFLOPS = 78.6   # fp64 matrix = fp64 vector rate on gfx950 (public spec)


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


def make_landmarks(x, m, how, ctx, seed=42, sub=100_000):
    """k-means centroids of the first <= 1e5 cells, random_state 42, computed once, untimed.
    how = "sklearn": sklearn.cluster.k_means(X[:1e5], m, n_init=1, random_state=42), the reference's own call
    (parameters.py:291); "device": the library's k-means++ / Lloyd (mln_kmeans) on the same subsample."""
    xs = x[:min(sub, x.shape[0])]
    if m >= xs.shape[0]:
        return np.ascontiguousarray(xs[:m]), "first m cells"
    t0 = time.perf_counter()
    if how == "sklearn":
        from sklearn.cluster import k_means
        c = k_means(xs, m, n_init=1, random_state=seed)[0]
        what = f"sklearn.cluster.k_means(X[:{xs.shape[0]}], {m}, n_init=1, random_state={seed})"
    else:
        c = ctx.kmeans(xs, m, seed=seed)
        what = f"mln_kmeans (k-means++ / Lloyd on the device) on X[:{xs.shape[0]}], seed {seed}"
    dt = time.perf_counter() - t0
    # centroids differ in their last bits from process to process (threaded BLAS / OpenMP reductions), enough to move
    # the L-BFGS pass count by a few evaluations; rounded through float32 they are the same numbers in every run
    c = np.ascontiguousarray(c.astype(np.float32).astype(np.float64))
    return c, f"{what}, rounded through float32, host, untimed ({dt:.1f} s)"


def cpu_baseline(x, landmarks, nn, kern_name, samples, n_target):
    """The oracle (NumPy/SciPy restatement of the reference's JAX-CPU path, reference stopping rule) timed on this
    box's host cores on the first `s` cells of the SAME workload for each s in `samples`; every stage is O(n), so
    the per-cell time of the largest sample extrapolates linearly to the n_target cells of the GPU run
    (the full config needs >= 80 GB of host RAM and tens of minutes)."""
    from oracle import mellon_oracle as mo
    pts = []
    for s in samples:
        xs, nns = x[:s], nn[:s]
        t0 = time.perf_counter()
        fit = mo.density_fit(xs, cov_func_curry=getattr(mo, kern_name), landmarks=landmarks, nn_distances=nns)
        dt = time.perf_counter() - t0
        pts.append({"cells": int(s), "seconds": round(dt, 2), "objective_evaluations": int(fit.n_eval)})
        del fit
        gc.collect()
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    big = pts[-1]
    value = big["cells"] / big["seconds"]
    extrap = None
    if len(pts) >= 2 and pts[-1]["cells"] != pts[0]["cells"]:
        # linear model t = a + b n through the two largest samples
        (n0, t0_), (n1, t1_) = [(p["cells"], p["seconds"]) for p in pts[-2:]]
        b = (t1_ - t0_) / (n1 - n0)
        a = t1_ - b * n1
        extrap = a + b * n_target
    return {"value": value, "unit": "cells/s", "cores": int(threads), "kind": "port",
            "sample": "first s cells of the workload for s in " + str([p["cells"] for p in pts])
                      + f", m={landmarks.shape[0]}, d={x.shape[1]}, reference L-BFGS-B defaults, "
                      f"os.cpu_count()={os.cpu_count()}; value = cells / wall of the largest sample",
            "points": pts,
            "extrapolated_seconds_at_full_size": None if extrap is None else round(extrap, 1),
            "extrapolated_cells_per_s_at_full_size": None if not extrap else n_target / extrap}


def _stdout_to_stderr():
    """Route file descriptor 1 to stderr and return a handle on the real stdout.  Libraries loaded by the run print
    to the C-level stdout on their own (RCCL's version banner at communicator set-up, which C stdio only flushes at
    exit, i.e. AFTER Python's output); the contract is ONE JSON line there."""
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



def roofline_objects(stats, step_s, ach, per_launch, n64, traffic, traffic_src, ach32, per_launch32, n32, bytes32, traffic32):
    """`roofline` describes the DOMINANT kernel of the step -- since the MAP solve runs all but two of its passes on the
    32-bit copy that is k_objective32; the fp64 kernel (same structure, 8 B/element) is reported beside it."""
    r64 = {"bound": "hbm", "kernel": "k_objective (fused loss+grad, one pass over the fp64 buffer: the anchor and the "
           "verification of the MAP solve)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
           "algorithmic_bytes_per_launch": stats["objective_bytes_per_launch"], "avg_launch_ms": 1e3 * per_launch,
           "launches": n64, "share_of_step": stats["objective_kernel_s"] / step_s if step_s else None}
    r32 = {"bound": "hbm", "kernel": "k_objective32 (fused loss+grad, one pass over the 32-bit fixed-point copy of K: "
           "every other pass of the MAP solve)", "achieved": ach32, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": ach32 / HBM_PEAK_GBS, "traffic": traffic32, "traffic_source": traffic_src,
           "algorithmic_bytes_per_launch": bytes32, "avg_launch_ms": 1e3 * per_launch32, "launches": n32,
           "share_of_step": stats.get("objective32_kernel_s", 0.0) / step_s if step_s else None}
    if stats.get("objective32_kernel_s", 0.0) > stats["objective_kernel_s"]:
        return {"roofline": r32, "roofline_fp64_passes": r64}
    return {"roofline": r64, "roofline_fp32_passes": r32}

def main():
    real_stdout = _stdout_to_stderr()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", dest="n", type=int, default=1_000_000, help="cells in total (strong) / per GPU (weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--dims", dest="d", type=int, default=50)
    ap.add_argument("--landmarks", dest="m", type=int, default=5000)
    ap.add_argument("--landmark-method", choices=["sklearn", "device"], default="sklearn")
    ap.add_argument("--kernel", default="Matern52")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=30000,
                    help="largest CPU-baseline sample in cells (a second point at half of it gives the slope); 0 = skip")
    ap.add_argument("--cpu-sample-full", action="store_true",
                    help="SURVEY S8(d) sizes: 2.5e5 and 5e5 cells (needs ~60 GB of host RAM and ~15 min)")
    ap.add_argument("--extra-steps", type=int, default=2, help="steps of the fp64-only and host-to-host measurements")
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
    # replicated inputs must be BIT-identical on every rank (they steer the shared optimiser): rank 0 computes them
    lm_pack = make_landmarks(x0, m, args.landmark_method, ctx) if rank == 0 else None
    landmarks, lm_note = comm.broadcast(lm_pack, src=0)
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
        x_loc = np.ascontiguousarray(x0[lo:hi])
        x_all_dev = ctx.to_device(x0)
        x_loc_dev = ctx.to_device(x_loc) if world > 1 else x_all_dev
        t0 = time.perf_counter()
        nn_loc = ctx.nn_distances(x_loc_dev, x_all_dev, self_offset=lo)      # exact 1-NN, excluded from timing
        t_nn = time.perf_counter() - t0
        if world > 1:
            x_all_dev.free()
    t_gen = time.perf_counter() - t_gen
    kern = getattr(mellon_amd.cov, args.kernel)

    def one_step(x_in):
        # check_rank=False: the rank diagnostic is log-only (skipped in timed runs, CPU baseline alike)
        est = mellon_amd.DensityEstimator(cov_func_curry=kern, landmarks=landmarks, nn_distances=nn_loc, check_rank=False)
        dens = est.fit_predict(x_in)
        return est, dens

    def fence():
        comm.barrier()
        ctx.synchronize()

    def release(est):
        """Return the fit's device buffers (to the library's cache) deterministically."""
        est._fit.close()

    def timed(steps, x_in, keep_last=False):
        """EXACTLY `steps` steps between two fences; MAX over ranks of the wall time."""
        gc.collect()
        fence()
        t_fit = t_free = 0.0
        last = None
        t0 = time.perf_counter()
        for i in range(steps):
            ta = time.perf_counter()
            est, dens = one_step(x_in)
            tb = time.perf_counter()
            stats = est._fit.stage_times()
            n_eval = est.loss_func.n_eval
            if i + 1 < steps or not keep_last:
                release(est)
                est = None
            t_fit += tb - ta
            t_free += time.perf_counter() - tb
            last = (est, dens, stats, n_eval)
        fence()
        elapsed = time.perf_counter() - t0
        elapsed = float(comm.allreduce_sum(np.eye(world)[rank] * elapsed).max())   # MAX over ranks
        return elapsed, last, t_fit, t_free

    for _ in range(args.warmup):
        est, dens = one_step(x_loc_dev)
        release(est)
        del est
    elapsed, (est, dens, stats, n_eval), t_fit, t_free = timed(args.steps, x_loc_dev, keep_last=True)

    # ---- size-independent parity property at full size: predict(X) == fit_predict(X) -----------------
    k = min(20000, hi - lo)
    xq = x_loc[:k]
    prop = float(np.abs(est.predict(xq) - dens[:k]).max() / np.abs(dens[:k]).max())
    release(est)
    del est

    # ---- beside the headline, same process, same inputs (untimed by the contract) -------------------------
    extra = {}
    if args.extra_steps > 0:
        e_h2h, (_, dens_h, _, n_eval_h), _, _ = timed(args.extra_steps, x_loc)          # host x -> host density
        extra["ms_per_step_host_to_host"] = 1e3 * e_h2h / args.extra_steps
        os.environ["MELLON_AMD_MIXED"] = "0"
        one_step(x_loc_dev)[0]._fit.close()                                             # allocator warm-up of the other buffer set
        e_f64, (_, dens64, stats64, n_eval64), _, _ = timed(args.extra_steps, x_loc_dev)
        del os.environ["MELLON_AMD_MIXED"]
        extra["ms_per_step_fp64_only"] = 1e3 * e_f64 / args.extra_steps
        extra["objective_evaluations_fp64_only"] = int(n_eval64)
        extra["fp64_only_vs_mixed_rel_max"] = float(np.abs(dens64 - dens).max() / np.abs(dens).max())

    if rank != 0:
        return
    ms_per_step = 1e3 * elapsed / args.steps
    value = n_total * args.steps / elapsed
    per_launch = stats["objective_kernel_s"] / max(stats["objective_launches"], 1.0)
    ach = stats["objective_bytes_per_launch"] / per_launch / 1e9 if per_launch > 0 else 0.0
    # fp32 warm-up passes of the MAP solve (mixed precision): same rows, 4 bytes per element
    n32 = stats.get("objective32_launches", 0.0)
    per_launch32 = stats.get("objective32_kernel_s", 0.0) / max(n32, 1.0)
    bytes32 = stats["objective_bytes_per_launch"] / 2.0
    ach32 = bytes32 / per_launch32 / 1e9 if n32 > 0 and per_launch32 > 0 else 0.0
    n64 = int(stats["objective_launches"])
    traffic = traffic32 = None
    traffic_src = None
    tfile = os.path.join(ROOT, "profiles", "objective_traffic.json")
    if os.path.exists(tfile):
        try:
            t = json.load(open(tfile))
            if t.get("n_local") == hi - lo and t.get("m") == m:
                traffic = t.get("hbm_bytes_per_launch")
                traffic32 = t.get("fp32_passes", {}).get("hbm_bytes_per_launch")
                traffic_src = "profiles/objective_traffic.json (rocprofv3 PMC passes of this command, committed; not re-measured in this run)"
        except Exception:
            traffic = traffic32 = None
    out = {
        "metric": "cells/sec fit_predict", "value": value, "unit": "cells/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None,
        "dtype": (f"f64 results; {int(n32)} of {int(n_eval)} passes stream a 32-bit copy of K (4 B/element: "
                  + ("fixed point round(K 2^32)" if stats.get("copy32_format") == 2.0 else "fp32")
                  + f"), {n64} the fp64 buffer (they anchor the first-order correction of the 32-bit objective and verify "
                    "the final point: loss, gradient and log-density of the returned optimum are fp64 evaluations) -- "
                    "pure-fp64 step: ms_per_step_fp64_only") if n32 > 0 else "f64",
        "data": "synthetic",
        **extra,
        "config": {"workload": f"C3 DensityEstimator.fit_predict: {n_total} cells x {d} dims Gaussian mixture "
                               f"(seed {args.seed}), {m} landmarks, {args.kernel}, one model, cells sharded over "
                               f"{world} GPU(s) ({hi - lo} cells per GPU)",
                   "n": n_total, "n_per_gpu": hi - lo, "d": d, "m": m, "kernel": args.kernel,
                   "parallelism": f"cells/{world}",
                   "objective_evaluations": int(n_eval), "objective_evaluations_32bit": int(n32),
                   "objective_evaluations_fp32": int(n32),   # (same number under its round-1 name)
                   "optimizer": "device-resident L-BFGS maxcor=10 ftol=1e-13 gtol=1e-7 on a preconditioned variable; path "
                                "shortcuts that leave the optimum alone (DESIGN.md S4): capped start, step-length memory, "
                                "first-order-corrected 32-bit surrogate verified by an fp64 evaluation of the final point",
                   "landmarks": lm_note,
                   "nn_distances": f"exact 1-NN on device, untimed ({t_nn:.2f} s)",
                   "timed_region": "x, landmarks, nn_distances resident (x in HBM) -> log-density in host memory; "
                                   "ms_per_step_host_to_host starts from x in host memory (BASELINE.md S2)",
                   "predict_equals_fit_predict_rel_max": prop, "device": info["arch"]},
        **roofline_objects(stats, elapsed / args.steps if world == 1 else None, ach, per_launch, n64, traffic, traffic_src,
                           ach32, per_launch32, int(n32), bytes32, traffic32),
        "stages_s": {k: round(v, 4) for k, v in stats.items() if k.endswith("_s")},
        "host_s": {"fit_predict_per_step": round(t_fit / args.steps, 4), "release_per_step": round(t_free / args.steps, 4)},
    }
    mfile = os.path.join(ROOT, "profiles", "mfma_util.json")
    if os.path.exists(mfile):
        try:
            out["mfma"] = json.load(open(mfile))
        except Exception:
            pass
    if world == 1 and (args.cpu_sample > 0 or args.cpu_sample_full):
        gc.collect()
        if args.cpu_sample_full:
            samples = [250_000, 500_000]
        else:
            big = min(args.cpu_sample, n)
            samples = [max(big // 2, 1), big]
        out["cpu_baseline"] = cpu_baseline(x0, landmarks, nn_loc, args.kernel, samples, n_total)
    _print_result_line(real_stdout, json.dumps(out))


if __name__ == "__main__":
    main()
