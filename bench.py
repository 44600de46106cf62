#!/usr/bin/env python
"""bench.py -- cells/sec of DensityEstimator.fit_predict on MI355X (BASELINE.json metric).

One "step" = one complete fit_predict pass of the hot path over the synthetic workload
(covariance tiles -> Cholesky -> Ridge init / preconditioner -> L-BFGS MAP solve on the fused device
objective -> log-density), inputs already resident in HBM when the timed region starts.
Workload: BASELINE config 3 -- 1e6 cells x 50 dims Gaussian mixture, 5 000 landmarks, Matern52.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N > 1: ONE model, its cells sharded over the N ranks -- one process per GPU, RCCL all-reduce of (loss, grad) per evaluation
and of the preconditioner's Gram once per build; no collective touches the n x m buffer.  Default `"scaling": "strong"`: the
1e6 cells of BASELINE config 3 IN TOTAL, split N ways -- at every N the workload is the one BASELINE.json names and north_star's
">= 6x at 8 GPUs" is quoted on; `value` = 1e6 cells x steps / time.  The WEAK step (1e6 cells PER GPU, one model on N x 1e6
cells) is measured in the same run and reported beside it under `weak_scaling`; `--scaling weak` makes it the headline instead.
Only the launcher's environment variables (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) are used; the ranks' host sides talk
over a Unix socket.

Prints ONE JSON line on rank 0 (fields documented in DESIGN.md S6).  The K timed steps are PURE FLOAT64: every pass
of the MAP solve streams the fp64 n x m buffer (MELLON_AMD_MIXED=0), so `value`, `ms_per_step`, `roofline` and
`dtype` describe the reference's own precision.  Beside it, untimed by the contract but in the same process: the
product default with its 32-bit fixed-point surrogate passes (`ms_per_step_mixed`, `roofline_mixed_passes`) and the
host-to-host fp64 step (`ms_per_step_host_to_host`, x uploaded inside the step).

    --config c2 | c4 | c5      the other BASELINE configs as bench lines (same contract, their own workloads)
    --dry-run-comm             only set up the communicator, run its self-test and one all-reduce, print a JSON line
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
PREDICT_EPILOGUE_FLOPS = 14.0  # c_k of SURVEY S8(d)'s predict formula: 12 + 1 exp + 1 sqrt per kernel-matrix element


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


def cpu_baseline_predict(x, landmarks, state, kern_name, gpu_pred, sample=60_000, block=10_000):
    """The oracle's Predictor (mu + cov(Xnew, landmarks) w: conditional.py:899-906 restated in NumPy) timed on the first
    `sample` cells with the weights of the device fit; also the parity of the device prediction on those rows."""
    from oracle import mellon_oracle as mo
    ls, w, mu = state
    s = min(sample, x.shape[0])
    pred = mo.Predictor(getattr(mo, kern_name)(ls), landmarks, w, mu, x.shape[0])
    out = np.empty(s)
    t0 = time.perf_counter()
    for i0 in range(0, s, block):
        out[i0:i0 + block] = pred(x[i0:i0 + block])
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        threads = max([p_.get("num_threads", 1) for p_ in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    return {"value": s / dt, "unit": "cells/s", "cores": int(threads), "kind": "port",
            "sample": f"oracle Predictor.__call__ on the first {s} cells in blocks of {block}, m={landmarks.shape[0]}, "
                      f"d={x.shape[1]}, weights of the device fit, os.cpu_count()={os.cpu_count()}",
            "seconds": round(dt, 2),
            "device_vs_oracle_rel_max": float(np.abs(gpu_pred[:s] - out).max() / np.abs(out).max())}


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



def roofline_object(kernel, bytes_per_launch, kernel_s, launches, step_s, traffic, traffic_src):
    per_launch = kernel_s / max(launches, 1.0)
    ach = bytes_per_launch / per_launch / 1e9 if per_launch > 0 and launches > 0 else 0.0
    return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": 1e3 * per_launch,
            "launches": int(launches), "share_of_step": (kernel_s / step_s) if step_s else None}


def committed_traffic(n_local, m):
    """HBM bytes per launch of the two objective kernels from the committed rocprofv3 PMC passes (separate --pmc
    FETCH_SIZE / WRITE_SIZE runs with --kernel-trace only; gfx950 correction x2 on FETCH_SIZE), with the commit the
    library was at when they were taken."""
    tfile = os.path.join(ROOT, "profiles", "objective_traffic.json")
    try:
        t = json.load(open(tfile))
    except Exception:
        return None, None, None
    if t.get("n_local") != n_local or t.get("m") != m:
        return None, None, None
    src = ("profiles/objective_traffic.json (rocprofv3 PMC passes of this command, committed; measured at commit "
           + str(t.get("measured_at_commit", "unrecorded")) + ", kernel source hash "
           + str(t.get("objective_hip_sha16", "unrecorded")) + "; not re-measured in this run)")
    return t.get("hbm_bytes_per_launch"), t.get("fp32_passes", {}).get("hbm_bytes_per_launch"), src


def measure_traffic_pmc(args):
    """--pmc: HBM bytes per full-size launch of the dominant kernel re-measured NOW: two child runs of one fp64 step of
    the same workload under `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, no
    other trace domain -- guides/MI355X_MICROARCH.md 'HBM'), gfx950 correction FETCH_SIZE x 2 (KB units).  Returns
    (bytes per launch, description) or (None, why not)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found on this box"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="mln_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", tmp, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
               "--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--extra-steps", "0", "--landmark-method", "device",
               "--cells", str(args.n), "--dims", str(args.d), "--landmarks", str(args.m), "--kernel", args.kernel,
               "--seed", str(args.seed)]
        env = dict(os.environ, TMPDIR="/tmp")
        env["MELLON_AMD_MIXED"] = "0"
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
            dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {counter} child failed (rc {r.returncode}): {r.stderr[-300:]}"
            con = sqlite3.connect(dbs[0])
            rows = con.execute("select kernel_name, value, duration from counters_collection where counter_name = ? "
                               "and kernel_name like '%k_objective<%'", (counter,)).fetchall()
            if not rows:
                return None, f"no k_objective rows in the {counter} pass"
            dmax = max(r_[2] for r_ in rows)
            full = [r_ for r_ in rows if r_[2] > 0.5 * dmax]          # the full-size launches (not gated off, not strided)
            got[counter] = (sum(r_[1] for r_ in full) / len(full), len(full), sum(r_[2] for r_ in full) / len(full) / 1e3)
        except Exception as e:          # noqa: BLE001 -- a profiling problem must not lose the bench line
            return None, f"rocprofv3 --pmc {counter}: {e!r}"
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    fetch_kb, nf, us_f = got["FETCH_SIZE"]
    write_kb, nw, us_w = got["WRITE_SIZE"]
    return 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0, (
        f"measured in THIS run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate child passes of one fp64 step "
        f"(bench.py --steps 1 --warmup 0 --landmark-method device); mean over the {nf} / {nw} full-size k_objective launches "
        f"({us_f:.0f} / {us_w:.0f} us under the profiler); FETCH_SIZE {fetch_kb:.0f} KB x 2 (gfx950 correction) + WRITE_SIZE "
        f"{write_kb:.0f} KB")


def c4_workload(n, d, T, seed):
    """BASELINE C4: Gaussian-mixture cells at T equally sized time points whose component means drift
    linearly in time; the time column is appended.  Rows are ordered by time."""
    xs = gaussian_mixture(n, d, seed)
    times = np.repeat(np.arange(float(T)), n // T)
    times = np.concatenate([times, np.full(n - times.size, float(T - 1))])
    xs = xs + 0.2 * times[:, None]
    return np.ascontiguousarray(np.concatenate([xs, times[:, None]], axis=1))


def bench_other(args, comm, ctx, info, world, rank):
    """C4 (TimeSensitiveDensityEstimator.fit_predict) and C5 (FunctionEstimator fit + batched predict) under the same
    contract as the headline: resident inputs, W untimed + K timed steps between fences, MAX over ranks, cells sharded
    over the ranks in contiguous blocks (replicated landmarks; one model)."""
    import mellon_amd
    from mellon_amd import distributed
    n, d, m = args.n, args.d, args.m
    kern = getattr(mellon_amd.cov, args.kernel)
    os.environ["MELLON_AMD_MIXED"] = "0"          # float64 throughout, like the headline
    lo, hi = distributed.shard_bounds(n, world, rank)

    def fence():
        comm.barrier()
        ctx.synchronize()

    if args.config == "c4":
        T, ls_time = 8, 1.5
        xt = c4_workload(n, d, T, args.seed)
        x_loc = np.ascontiguousarray(xt[lo:hi])
        t0 = time.perf_counter()
        # exact 1-NN within each time point, this rank's cells against the time point's cells of all ranks (untimed)
        from mellon_amd.parameters import compute_nn_distances_within_time_points
        nn_loc = compute_nn_distances_within_time_points(xt, local=(lo, hi - lo))
        t_nn = time.perf_counter() - t0
        ls = float(np.exp(comm.global_mean(np.log(nn_loc)) + 3.0))
        if rank == 0:                               # k-means of a 20 000-cell subsample with the time column rescaled (parameters.py:294-349)
            rng = np.random.default_rng(args.seed)
            km = xt[rng.choice(n, min(n, 20000), replace=False)].copy()
            km[:, -1] *= ls / ls_time
            lm = ctx.kmeans(km, m, seed=42) if m < km.shape[0] else km[:m]
            lm[:, -1] /= ls / ls_time
            lm = np.ascontiguousarray(lm.astype(np.float32).astype(np.float64))
        else:
            lm = None
        lm = comm.broadcast(lm, src=0)

        def one_step():
            est = mellon_amd.TimeSensitiveDensityEstimator(cov_func_curry=kern, landmarks=lm, nn_distances=nn_loc,
                                                           ls_time=ls_time, d=d, check_rank=False)
            dens = est.fit_predict(x_loc)
            return est, dens

        what = (f"C4 TimeSensitiveDensityEstimator.fit_predict: {n} cells x {d} dims (+ time column) at {T} time points, "
                f"{m} landmarks, {args.kernel}(ls) * {args.kernel}(ls_time=1.5), one model, cells sharded over {world} GPU(s)")
        unit_cells = n
    else:
        p, sigma = 2000, 0.1
        x = gaussian_mixture(n, d, args.seed)
        rng = np.random.default_rng(args.seed)
        Wm = rng.normal(size=(d, p)) / np.sqrt(d)
        x_loc = np.ascontiguousarray(x[lo:hi])
        y_loc = np.sin(x_loc @ Wm) + 0.1 * np.random.default_rng([args.seed, rank]).normal(size=(hi - lo, p))
        t0 = time.perf_counter()
        x_all_dev = ctx.to_device(x)
        nn_loc = ctx.nn_distances(ctx.to_device(x_loc) if world > 1 else x_all_dev, x_all_dev, self_offset=lo)
        t_nn = time.perf_counter() - t0
        x_all_dev.free()
        lm_pack = make_landmarks(x, m, "device", ctx) if rank == 0 else None
        lm = comm.broadcast(lm_pack, src=0)[0]

        # resident inputs AND outputs (the contract's timed region): cells, targets and the n x p predictions live in HBM
        x_dev, y_dev = ctx.to_device(x_loc), ctx.to_device(y_loc)
        out_dev = ctx.empty((hi - lo, p))

        def one_step():
            est = mellon_amd.FunctionEstimator(cov_func_curry=kern, sigma=sigma, landmarks=lm, nn_distances=nn_loc)
            est.fit(x_dev, y_dev)
            pred = est.predict(x_dev, out=out_dev)    # batched predict of this rank's cells, all p outputs
            return est, pred

        what = (f"C5 FunctionEstimator.fit + batched predict: {n} cells x {d} dims, {p} outputs (sin(X W) + noise), sigma 0.1, "
                f"{m} landmarks, {args.kernel}, Xnew = X, cells sharded over {world} GPU(s); x, y and the predictions resident in HBM")
        unit_cells = n

    for _ in range(args.warmup):
        est, _ = one_step()
        del est
    gc.collect()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        est, res = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = float(comm.allreduce_sum(np.eye(world)[rank] * elapsed).max())
    out = {"metric": "cells/sec fit_predict", "value": unit_cells * args.steps / elapsed, "unit": "cells/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": what, "n": n, "n_per_gpu": hi - lo, "d": d, "m": m, "kernel": args.kernel,
                      "parallelism": f"cells/{world}", "nn_distances": f"exact 1-NN on device, untimed ({t_nn:.2f} s)",
                      "device": info["arch"]}}
    if args.config == "c4":
        st = est._fit.stage_times()
        out["config"]["objective_evaluations"] = int(est.loss_func.n_eval)
        out["config"]["objective_full_pass_equivalents"] = st.get("objective_pass_equivalents")
        out["roofline"] = roofline_object("k_objective (fp64, one pass over the n x m buffer per evaluation)",
                                          st["objective_bytes_per_launch"], st["objective_kernel_s"], st["objective_launches"],
                                          elapsed / args.steps if world == 1 else None, None, None)
        out["stages_s"] = {k: round(v, 4) for k, v in st.items() if k.endswith("_s")}
    else:
        # the two GEMM-shaped parts: A A^T (n m^2) + A r (n m p) in the fit, K W (n m p) in the predict
        flops = (hi - lo) * (2.0 * m * m + 4.0 * m * 2000)
        out["roofline"] = {"bound": "mfma", "kernel": "fp64 MFMA GEMMs of the landmark conditional (A A^T, A r) and of the "
                           "batched predict (K W), whole step", "achieved": flops / (elapsed / args.steps) / 1e12,
                           "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": flops / (elapsed / args.steps) / 1e12 / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                           "note": "step-level figure (host<->device traffic of the n x p arrays included)"}
    # ---- CPU baseline beside it (rank 0, bounded sample of the SAME workload, the oracle = the reference's algorithm restated;
    #      kind "port"): C4 a time-sensitive density fit of an evenly spaced subsample, C5 the function fit + predict of one ------
    if rank == 0 and args.cpu_sample != 0:
        from oracle import mellon_oracle as mo
        try:
            from threadpoolctl import threadpool_limits
            limits = threadpool_limits(limits=os.cpu_count())
        except Exception:
            limits = None
        t0 = time.perf_counter()
        if args.config == "c4":
            ns = min(n, args.cpu_sample if args.cpu_sample > 0 else 40_000)
            idx = np.arange(0, n, max(1, n // ns))[:ns]
            xs = np.ascontiguousarray(xt[idx])
            nn_s = mo.per_time_nn_distances(xs[:, :-1], xs[:, -1])
            fit = mo.density_fit(xs, cov_func_curry=getattr(mo, args.kernel), landmarks=lm, nn_distances=nn_s, d=d, ls_time=1.5)
            evals = int(getattr(fit, "n_eval", 0) or 0)
            what_cpu = (f"{ns} cells (every {max(1, n // ns)}-th, all 8 time points), {m} landmarks, the oracle's time-sensitive density fit at the "
                        f"reference's L-BFGS-B defaults ({evals} evaluations), 1-NN distances included")
        else:
            ns = min(n, args.cpu_sample if args.cpu_sample > 0 else 20_000)
            idx = np.arange(0, n, max(1, n // ns))[:ns]
            xs = np.ascontiguousarray(x[idx])
            ys = np.sin(xs @ Wm) + 0.1 * np.random.default_rng(args.seed + 1).normal(size=(ns, p))
            nn_s = mo.exact_nn_distances(xs)
            pred = mo.function_fit(xs, ys, sigma, cov_func_curry=getattr(mo, args.kernel), landmarks=lm, nn_distances=nn_s)
            _ = pred.mean(xs)
            what_cpu = (f"{ns} cells (every {max(1, n // ns)}-th), {p} outputs, {m} landmarks: the oracle's landmark conditional + predict of "
                        "the same cells, 1-NN distances included")
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": ns / dt, "unit": "cells/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": what_cpu + f"; {dt:.1f} s wall (every stage is O(n): cells/s carries over to the full size)"}
    return out


def pick_cpu_samples(args, n, host_gb):
    """the CPU baseline belongs at 2.5e5 (and 5e5) cells.  The default run takes the 2.5e5-cell point when
    the host has the memory for it (L + its transposed copy + the Ridge Gram: ~25 GB at m = 5000) AND a quick probe
    predicts that it fits the time budget; otherwise the largest sample that does, saying so."""
    if args.cpu_sample_full:
        return [250_000, 500_000], "SURVEY S8(d) sizes (--cpu-sample-full)"
    if args.cpu_sample > 0:
        big = min(args.cpu_sample, n)
        return [max(big // 2, 1), big], f"--cpu-sample {args.cpu_sample}"
    return None, None


def cpu_baseline_budgeted(x, landmarks, nn, kern_name, n_target, budget_s, host_gb):
    """Probe at 12 000 cells, then the largest of (2.5e5, 1.25e5, 6e4, 3e4) cells whose predicted time (linear in n:
    every stage is O(n) at fixed m, and the evaluation count is n-independent within ~10 %) fits `budget_s` and whose
    memory fits the host."""
    probe = cpu_baseline(x, landmarks, nn, kern_name, [12_000], n_target)
    per_cell = probe["points"][0]["seconds"] / 12_000.0
    m = landmarks.shape[0]
    note = []
    for cand in (250_000, 125_000, 60_000, 30_000):
        if cand > x.shape[0]:
            continue
        need_gb = 3.2 * cand * m * 8 / 1e9          # L, the transposed copy of decomposition.py:209, solver temporaries
        pred = per_cell * cand
        if need_gb > 0.8 * host_gb:
            note.append(f"{cand} cells need ~{need_gb:.0f} GB of host RAM ({host_gb:.0f} GB here)")
            continue
        if pred > budget_s:
            note.append(f"{cand} cells predicted {pred:.0f} s > budget {budget_s:.0f} s")
            continue
        out = cpu_baseline(x, landmarks, nn, kern_name, [cand], n_target)
        out["points"] = probe["points"] + out["points"]
        (n0, t0_), (n1, t1_) = [(q["cells"], q["seconds"]) for q in out["points"][-2:]]
        b = (t1_ - t0_) / (n1 - n0)
        extrap = (t1_ - b * n1) + b * n_target
        out["extrapolated_seconds_at_full_size"] = round(extrap, 1)
        out["extrapolated_cells_per_s_at_full_size"] = n_target / extrap if extrap > 0 else None
        out["sample"] += "; probe at 12000 cells first" + ("; skipped: " + "; ".join(note) if note else "")
        out["time_budget_s"] = budget_s
        return out
    probe["sample"] += "; no larger sample fits the budget: " + "; ".join(note)
    return probe


def host_memory_gb():
    try:
        return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 1e9
    except (ValueError, OSError):
        return 0.0


def main():
    real_stdout = _stdout_to_stderr()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["c3", "c2", "c4", "c5"], default="c3",
                    help="BASELINE.json config: c3 = the headline (1e6 x 50, 5000 landmarks, Matern52); c2 / c4 / c5 time "
                         "the other configs under the same contract")
    ap.add_argument("--cells", dest="n", type=int, default=None, help="cells in total (strong) / per GPU (weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): --cells in total, split N ways -- at every N the workload is BASELINE config 3 itself "
                         "(1e6 cells), which is what north_star's '>= 6x at 8 GPUs' is quoted on; weak: --cells per GPU, one model "
                         "on N x cells.  With N > 1 the other mode's step is measured too (--extra-steps) and reported beside the headline")
    ap.add_argument("--dims", dest="d", type=int, default=None)
    ap.add_argument("--landmarks", dest="m", type=int, default=None)
    ap.add_argument("--landmark-method", choices=["sklearn", "device"], default="sklearn")
    ap.add_argument("--kernel", default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="largest CPU-baseline sample in cells (a second point at half of it gives the slope); 0 = skip; "
                         "default: 2.5e5 cells when the host's memory and --cpu-budget-s allow, else the largest that fits")
    ap.add_argument("--cpu-budget-s", type=float, default=330.0, help="wall-clock budget of the default CPU baseline")
    ap.add_argument("--cpu-sample-full", action="store_true",
                    help="SURVEY S8(d) sizes: 2.5e5 and 5e5 cells (needs ~60 GB of host RAM and ~15 min)")
    ap.add_argument("--extra-steps", type=int, default=4, help="steps of the mixed-precision and host-to-host measurements")
    ap.add_argument("--pmc", action="store_true",
                    help="re-measure roofline.traffic in this run: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) of one "
                         "fp64 step, ~1 min each; without it the committed profiles/objective_traffic.json is quoted")
    ap.add_argument("--allow-host-staged", action="store_true",
                    help="with --gpus N > 1: accept host-staged device collectives (RCCL unavailable, or ranks sharing a GPU); "
                         "without it a multi-rank run whose transport is not RCCL over N ranks exits non-zero")
    ap.add_argument("--dry-run-comm", action="store_true",
                    help="communicator set-up + self-test + one all-reduce of the per-evaluation size, nothing else")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
        args.gpus = world

    import mellon_amd
    from mellon_amd import _lib, distributed
    t_comm = time.perf_counter()
    comm = distributed.init_from_env()
    t_comm = time.perf_counter() - t_comm
    ctx = _lib.default_context()
    info = ctx.device_info()

    # ---- which transport carries the device collectives: said in the result line, and enforced -------------------------
    # A SCALE line must prove that RCCL saw N ranks (round-3 verdict): the transport's OWN report (ncclCommCount /
    # ncclCommUserRank), not what this process was told.  Anything else than RCCL over `world` ranks is a failed
    # multi-GPU run unless --allow-host-staged (or MELLON_AMD_SHARE_GPU=1, the documented several-ranks-per-GPU test mode).
    cinfo = ctx.comm_info()
    comm_report = {"comm_backend": getattr(comm, "backend", "none" if world == 1 else cinfo["transport"]),
                   "backend_note": getattr(comm, "backend_note", None),
                   "transport": cinfo["transport"], "ranks_reported_by_transport": cinfo["ranks_reported_by_transport"],
                   "rccl_version_code": cinfo["rccl_version_code"], "init_s": round(t_comm, 3)}
    if world > 1:
        all_reports = comm.host.allgather((cinfo["transport"], cinfo["ranks_reported_by_transport"],
                                           cinfo["rank_reported_by_transport"]))
        comm_report["per_rank_transport"] = [list(r) for r in all_reports]
        rccl_ok = all(r[0] == "rccl" and r[1] == world and r[2] == i for i, r in enumerate(all_reports))
        comm_report["rccl_over_all_ranks"] = rccl_ok
        if not rccl_ok and not (args.allow_host_staged or os.environ.get("MELLON_AMD_SHARE_GPU") == "1"):
            if rank == 0:
                print(f"[bench] --gpus {world}: the device collectives do not run over RCCL with {world} ranks "
                      f"({comm_report}); refusing to produce a scaling number (pass --allow-host-staged to accept)",
                      file=sys.stderr)
            comm.barrier()
            sys.exit(3)

    if args.dry_run_comm:
        rep = getattr(comm, "self_test_report", {"world_size": world, "ok": True})
        buf = np.full(5001, float(rank + 1))
        t0 = time.perf_counter()
        for _ in range(20):
            out_v = ctx.allreduce_sum(buf)
        dt = (time.perf_counter() - t0) / 20
        ok = bool(np.all(out_v == world * (world + 1) / 2.0))
        comm.barrier()
        if rank == 0:
            _print_result_line(real_stdout, json.dumps({
                "metric": "communicator dry run", "value": 1.0 if ok and rep.get("ok") else 0.0, "unit": "ok", "n_gpus": world,
                "steps": 0, "warmup": 0, "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": args.scaling,
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "communicator set-up, self-test (distributed.self_test) and 20 all-reduces of m + 1 = "
                                       "5001 fp64 from host buffers", "device": info["arch"]},
                "init_s": t_comm, "self_test": rep, "comm": comm_report}))
        return

    defaults = {"c3": (1_000_000, 50, 5000, "Matern52", 3), "c2": (100_000, 20, 1000, "ExpQuad", 2),
                "c4": (500_000, 30, 2000, "Matern52", 4), "c5": (200_000, 50, 2000, "Matern52", 5)}[args.config]
    args.n = args.n or defaults[0]
    args.d = args.d or defaults[1]
    args.m = args.m or defaults[2]
    args.kernel = args.kernel or defaults[3]
    args.seed = defaults[4] if args.seed is None else args.seed
    if args.config in ("c4", "c5"):
        out = bench_other(args, comm, ctx, info, world, rank)
        out["config"].update(comm_report)
        if rank == 0:
            _print_result_line(real_stdout, json.dumps(out))
        return

    # ---- synthetic workload -----------------------------------------------------------------------------
    n, d, m = args.n, args.d, args.m
    weak = args.scaling == "weak"
    t_gen = time.perf_counter()
    x0 = gaussian_mixture(n, d, args.seed)                # shard 0 == the BASELINE data set
    # replicated inputs must be BIT-identical on every rank (they steer the shared optimiser): rank 0 computes them
    lm_pack = make_landmarks(x0, m, args.landmark_method, ctx) if rank == 0 else None
    landmarks, lm_note = comm.broadcast(lm_pack, src=0)

    def prepare(weak_mode):
        """(x_loc host, x_loc in HBM, exact 1-NN distances of this rank's cells, cells in total, [lo, hi), seconds in the search)"""
        t_search = 0.0
        if weak_mode:
            # rank r owns shard r (n cells); exact 1-NN among ALL world * n cells: every rank regenerates the
            # other shards (deterministic streams, no communication) and keeps the running minimum
            xl = x0 if rank == 0 else gaussian_mixture(n, d, args.seed, shard=rank)
            xl_dev = ctx.to_device(xl)
            nnl = None
            for s_ in range(world):
                xs = xl if s_ == rank else (x0 if s_ == 0 else gaussian_mixture(n, d, args.seed, shard=s_))
                xs_dev = xl_dev if s_ == rank else ctx.to_device(xs)
                t0 = time.perf_counter()
                part = ctx.nn_distances(xl_dev, xs_dev, self_offset=0 if s_ == rank else -(n + 1))
                t_search += time.perf_counter() - t0
                nnl = part if nnl is None else np.minimum(nnl, part)
                if s_ != rank:
                    xs_dev.free()
                del xs
            return xl, xl_dev, nnl, n * world, 0, n, t_search
        lo_, hi_ = distributed.shard_bounds(n, world, rank)
        xl = np.ascontiguousarray(x0[lo_:hi_])
        x_all_dev = ctx.to_device(x0)
        xl_dev = ctx.to_device(xl) if world > 1 else x_all_dev
        t0 = time.perf_counter()
        nnl = ctx.nn_distances(xl_dev, x_all_dev, self_offset=lo_)      # exact 1-NN, excluded from timing
        t_search = time.perf_counter() - t0
        if world > 1:
            x_all_dev.free()
        return xl, xl_dev, nnl, n, lo_, hi_, t_search

    x_loc, x_loc_dev, nn_loc, n_total, lo, hi, t_nn = prepare(weak)
    t_gen = time.perf_counter() - t_gen
    kern = getattr(mellon_amd.cov, args.kernel)

    nn_cur = [nn_loc]          # (the strong-scaling companion measurement swaps in its own shard's distances)

    def one_step(x_in):
        # check_rank=False: the rank diagnostic is log-only (skipped in timed runs, CPU baseline alike)
        est = mellon_amd.DensityEstimator(cov_func_curry=kern, landmarks=landmarks, nn_distances=nn_cur[0], check_rank=False)
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

    # ---- the headline: PURE FLOAT64 (no 32-bit copy exists, every pass streams the fp64 buffer) ---------------
    os.environ["MELLON_AMD_MIXED"] = "0"
    for _ in range(args.warmup):
        est, dens = one_step(x_loc_dev)
        release(est)
        del est
    elapsed, (est, dens, stats, n_eval), t_fit, t_free = timed(args.steps, x_loc_dev, keep_last=True)

    # ---- size-independent parity property at full size: predict(X) == fit_predict(X) -----------------
    k = min(20000, hi - lo)
    xq = x_loc[:k]
    prop = float(np.abs(est.predict(xq) - dens[:k]).max() / np.abs(dens[:k]).max())

    # ---- the second half of SURVEY S8(d)'s metric: n' / wall(predict).  est.predict(X) for this rank's cells (resident in HBM;
    #      the log-density lands in host memory), landmarks and weights cached by the predictor: K(X, xu) is never materialised
    #      (conditional.py:899-906, base_predictor.py:180-257).  No collective: every rank predicts its own rows. -------------
    predict_line = predict_state = pd = None
    if args.extra_steps > 0:
        predictor = est.predict
        pd = predictor(x_loc_dev)                       # builds the predictor (weights w = Lp^-T z), warms the kernel
        predict_state = (float(est.ls), np.array(predictor.weights), float(predictor.mu))
        gc.collect()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.extra_steps):
            pd = predictor(x_loc_dev)
        fence()
        e_pr = time.perf_counter() - t0
        e_pr = float(comm.allreduce_sum(np.eye(world)[rank] * e_pr).max())
        flops_cell = (2.0 * d + PREDICT_EPILOGUE_FLOPS + 2.0) * m
        tf = flops_cell * (hi - lo) * args.extra_steps / e_pr / 1e12
        predict_line = {
            "metric": "cells/sec predict", "value": n_total * args.extra_steps / e_pr, "unit": "cells/s",
            "ms_per_call": 1e3 * e_pr / args.extra_steps, "steps": args.extra_steps, "cells_per_call_per_gpu": hi - lo,
            "region": "X resident in HBM -> log-density in host memory; predictor (landmarks, weights) built once, untimed",
            "equals_fit_predict_rel_max": float(np.abs(pd - dens).max() / np.abs(dens).max()),
            "roofline": {"bound": "mfma", "kernel": "k_predict_mean_rows (distances on the fp64 matrix cores, kernel epilogue and "
                         "the dot with the weights on the fp64 vector pipe -- ONE fp64 datapath per SIMD on gfx950, so the bound is "
                         "the fp64 pipe; whole call, per GPU)", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                         "algorithmic_flops_per_cell": flops_cell,
                         "formula": f"(2 d + c_k + 2) m per cell, c_k = {PREDICT_EPILOGUE_FLOPS:.0f} (SURVEY S8(d): 12 + 1 exp + 1 sqrt)"}}
    release(est)
    del est

    # ---- beside the headline, same process, same inputs (untimed by the contract) -------------------------
    extra = {}
    stats_mixed = None
    if args.extra_steps > 0:
        # BASELINE.md S2's region: validated X in HOST memory -> log-density in host memory.  The caller's array is page-locked
        # once (mln_host_register, what a data loader does; untimed, like the landmarks and nn distances) so that the upload is
        # a DMA transfer under the kernel-matrix pass; the same step from pageable memory is reported beside it.
        t_pin = time.perf_counter()
        with ctx.pinned(x_loc):
            extra["host_register_ms_once"] = 1e3 * (time.perf_counter() - t_pin)
            one_step(x_loc)[0]._fit.close()
            e_h2h, (_, dens_h, _, n_eval_h), _, _ = timed(args.extra_steps, x_loc)      # host x (pinned) -> host density, fp64
        extra["ms_per_step_host_to_host"] = 1e3 * e_h2h / args.extra_steps
        e_pg, _, _, _ = timed(args.extra_steps, x_loc)                                   # the same from pageable memory
        extra["ms_per_step_host_to_host_pageable"] = 1e3 * e_pg / args.extra_steps
        os.environ["MELLON_AMD_MIXED"] = "1"                                             # the opt-in mixed-precision solve (the product default is fp64 since round 5)
        one_step(x_loc_dev)[0]._fit.close()                                             # allocator warm-up of the other buffer set
        e_mx, (_, dens_mx, stats_mixed, n_eval_mx), _, _ = timed(args.extra_steps, x_loc_dev)
        extra["ms_per_step_mixed"] = 1e3 * e_mx / args.extra_steps
        extra["cells_per_s_mixed"] = n_total * args.extra_steps / e_mx
        extra["objective_evaluations_mixed"] = int(n_eval_mx)
        extra["objective_evaluations_mixed_32bit"] = int(stats_mixed.get("objective32_launches", 0.0))
        extra["mixed_vs_fp64_rel_max"] = float(np.abs(dens_mx - dens).max() / np.abs(dens).max())
        os.environ["MELLON_AMD_MIXED"] = "0"

    # ---- the other scaling mode beside the headline (N > 1): strong headline -> the weak step (1e6 cells PER GPU, one model on
    #      N x 1e6 cells) as the companion, and the other way round -------------------------------------------------------------
    if world > 1 and args.extra_steps > 0:
        _, xs_dev, nn_other, n_tot2, lo2, hi2, _ = prepare(not weak)
        nn_cur[0] = nn_other
        one_step(xs_dev)[0]._fit.close()                                                 # allocator warm-up at the new sizes
        e_st, (_, dens_st, stats_st, n_eval_st), _, _ = timed(args.extra_steps, xs_dev)
        extra["weak_scaling" if not weak else "strong_scaling"] = {
            "note": (f"{n} cells PER GPU (rank r owns an independent shard of the same mixture), one model on {n_tot2} cells; "
                     if not weak else
                     f"the same {n} cells (shard 0 of the weak run = the 1-GPU workload) split over {world} GPUs, one model; ")
                    + f"{args.extra_steps} fp64 steps between the same fences, MAX over ranks",
            "n": n_tot2, "n_per_gpu": hi2 - lo2, "ms_per_step": 1e3 * e_st / args.extra_steps,
            "value": n_tot2 * args.extra_steps / e_st, "unit": "cells/s", "objective_evaluations": int(n_eval_st),
            "precond_rebuilds": stats_st.get("precond_rebuilds")}
        xs_dev.free()
        nn_cur[0] = nn_loc

    # ---- the 1-GPU step of the SAME run: rank 0 alone, on a private single-rank context, fits the unsharded shard-0 workload
    #      (the BASELINE config-3 problem) while the other ranks wait -- so that the N > 1 line carries its own speed-up ----------
    if world > 1 and args.extra_steps > 0:
        n1_ms = None
        if rank == 0:
            solo = distributed.Communicator()
            solo.ctx = _lib.Context()
            distributed.set_thread_current(solo)
            try:
                x0d = solo.ctx.to_device(x0)
                nn_cur[0] = solo.ctx.nn_distances(x0d, x0d)
                one_step(x0d)[0]._fit.close()
                solo.ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.extra_steps):
                    one_step(x0d)[0]._fit.close()
                solo.ctx.synchronize()
                n1_ms = 1e3 * (time.perf_counter() - t0) / args.extra_steps
                x0d.free()
            finally:
                distributed.set_thread_current(None)
                nn_cur[0] = nn_loc
        comm.barrier()
        if rank == 0 and n1_ms is not None:
            extra["n1_ms_per_step_same_run"] = n1_ms
            extra["n1_note"] = f"{n} cells unsharded on rank 0's GPU alone, {args.extra_steps} fp64 steps, same process, same landmarks"
            if "strong_scaling" in extra:
                extra["strong_scaling"]["speedup_vs_n1"] = n1_ms / extra["strong_scaling"]["ms_per_step"]
            if "weak_scaling" in extra:
                extra["weak_scaling"]["throughput_vs_n1"] = extra["weak_scaling"]["value"] / (n / (1e-3 * n1_ms))
            if not weak:
                extra["speedup_vs_n1"] = n1_ms / (1e3 * elapsed / args.steps)

    # ---- per-rank cost of the collectives: one more fp64 step with a pair of stream events around every collective ------
    if world > 1:
        ctx.comm_info(timing=True, reset=True)
        e_c, _, _, _ = timed(1, x_loc_dev)
        ci = ctx.comm_info(timing=False, reset=True)
        keys = ("allreduce_calls", "small_allreduce_calls", "allreduce_bytes", "broadcast_calls", "allgather_calls",
                "large_allreduce_ms", "small_allreduce_ms", "broadcast_allgather_ms")
        per_rank = comm.host.allgather([ci[k] for k in keys])
        comm_report["collectives_per_step"] = {
            "note": "one extra fp64 step (untimed) with event pairs around every device collective; per rank, rank order",
            "step_ms_with_event_pairs": 1e3 * e_c, **{k: [r[i] for r in per_rank] for i, k in enumerate(keys)}}

    if rank != 0:
        return
    ms_per_step = 1e3 * elapsed / args.steps
    step_s = elapsed / args.steps
    value = n_total * args.steps / elapsed
    n64 = stats["objective_launches"]
    bytes64 = stats["objective_bytes_per_launch"]
    traffic, traffic32, traffic_src = committed_traffic(hi - lo, m)
    if args.pmc and world == 1:
        t_now, src_now = measure_traffic_pmc(args)
        if t_now is not None:
            traffic, traffic_src = t_now, src_now
        else:
            traffic_src = (traffic_src or "") + f" [--pmc failed: {src_now}]"
    roof = roofline_object("k_objective (fused loss + gradient, ONE pass over the fp64 n x m buffer per evaluation of the MAP "
                           "solve; also the Ridge right-hand side)", bytes64, stats["objective_kernel_s"], n64,
                           step_s if world == 1 else None, traffic, traffic_src)
    # step level, by SURVEY S8(d)'s byte formula with I = the evaluations actually performed (full-pass equivalents)
    passes = stats.get("objective_pass_equivalents", float(n_eval))
    step_bytes = 8.0 * (d + m * (passes + 3.0) + 1.0) * (hi - lo)
    roof["step_level"] = {"formula": "8 (d + m (I + 3) + 1) bytes per cell, I = objective passes in full-pass equivalents",
                          "I": passes, "bytes": step_bytes, "GBps": step_bytes / step_s / 1e9,
                          "frac_of_peak": step_bytes / step_s / 1e9 / HBM_PEAK_GBS}
    out = {
        "metric": "cells/sec fit_predict", "value": value, "unit": "cells/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        **extra,
        "config": {"workload": f"{args.config.upper()} DensityEstimator.fit_predict: {n_total} cells x {d} dims Gaussian mixture "
                               f"(seed {args.seed}), {m} landmarks, {args.kernel}, one model, cells sharded over "
                               f"{world} GPU(s) ({hi - lo} cells per GPU)",
                   "n": n_total, "n_per_gpu": hi - lo, "d": d, "m": m, "kernel": args.kernel,
                   "parallelism": f"cells/{world}",
                   "precision": "every kernel of the timed step computes in and streams float64 (MELLON_AMD_MIXED=0: the "
                                "32-bit fixed-point copy of the product default does not exist in this step)",
                   "objective_evaluations": int(n_eval), "objective_full_pass_equivalents": passes,
                   "optimizer": "device-resident L-BFGS maxcor=10 ftol=1e-13 gtol=1e-7 on a preconditioned variable; path "
                                "shortcuts that leave the optimum alone (DESIGN.md S4)",
                   "landmarks": lm_note,
                   "nn_distances": f"exact 1-NN on device, untimed ({t_nn:.2f} s)",
                   "timed_region": "x, landmarks, nn_distances resident (x in HBM) -> log-density in host memory; "
                                   "ms_per_step_host_to_host starts from x in host memory (BASELINE.md S2)",
                   "predict_equals_fit_predict_rel_max": prop, "device": info["arch"], **comm_report},
        "roofline": roof,
        "stages_s": {k: round(v, 4) for k, v in stats.items() if k.endswith("_s")},
        "host_s": {"fit_predict_per_step": round(t_fit / args.steps, 4), "release_per_step": round(t_free / args.steps, 4)},
    }
    if stats_mixed is not None and stats_mixed.get("objective32_launches", 0.0) > 0:
        out["roofline_mixed_passes"] = roofline_object(
            "k_objective32 (the same pass over the 32-bit fixed-point copy of K: the warm-up passes of the product default; "
            "NOT part of the timed fp64 step)", bytes64 / 2.0, stats_mixed["objective32_kernel_s"],
            stats_mixed["objective32_launches"], None, traffic32, traffic_src)
    mfile = os.path.join(ROOT, "profiles", "mfma_util.json")
    if os.path.exists(mfile):
        try:
            out["mfma"] = json.load(open(mfile))
        except Exception:
            pass
    if predict_line is not None:
        out["predict"] = predict_line
        if world == 1 and args.cpu_sample != 0:
            out["predict"]["cpu_baseline"] = cpu_baseline_predict(x0, landmarks, predict_state, args.kernel, pd)
    if world == 1 and args.cpu_sample != 0:
        gc.collect()
        samples, why = pick_cpu_samples(args, n, host_memory_gb())
        if samples is not None:
            out["cpu_baseline"] = cpu_baseline(x0, landmarks, nn_loc, args.kernel, samples, n_total)
            out["cpu_baseline"]["sample"] += "; " + why
        else:
            out["cpu_baseline"] = cpu_baseline_budgeted(x0, landmarks, nn_loc, args.kernel, n_total, args.cpu_budget_s,
                                                        host_memory_gb())
    _print_result_line(real_stdout, json.dumps(out))


if __name__ == "__main__":
    main()
