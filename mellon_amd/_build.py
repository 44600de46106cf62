"""Build libmellon_hip.so in-tree with hipcc for gfx950 (no GPU needed to compile).

Every source is compiled with -MMD, so that a later build recompiles exactly the translation units whose own
text or one of the headers they really include has changed (the persistent-row covariance kernels take minutes)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmellon_hip.so")
SOURCES = ["api.hip", "api_fit.hip", "api_precond.hip", "api_solve.hip", "api_noise.hip", "alloc.hip", "comm.hip", "cov_grad.hip", "cov_kernels.hip", "cov_rows.hip", "cov_rows_matern32.hip", "cov_rows_matern52.hip", "cov_rows_expquad.hip",
           "cov_rows_exponential.hip", "predict_rows.hip", "predict_rows_matern32.hip", "predict_rows_matern52.hip", "predict_rows_expquad.hip", "predict_rows_exponential.hip",
           "predict_rows_ratquad.hip", "predict_rows_prod.hip", "predict_rows_prod_matern32.hip", "predict_rows_prod_matern52.hip",
           "predict_rows_prod_expquad.hip", "predict_rows_prod_exponential.hip", "kernel_rows_prod_matern32.hip", "kernel_rows_prod_matern52.hip",
           "kernel_rows_prod_expquad.hip", "kernel_rows_prod_exponential.hip",
           "dgemm.hip", "diag.hip", "precond_rebuild.hip", "rowmin_f16.hip", "rowmin_w64.hip", "gram_i8.hip", "eigh.hip", "kmeans.hip", "linalg.hip", "potrf.hip", "objective.hip", "solver.hip", "tridiag.hip", "ldl_inertia.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + (["-DMLN_POTRF_TIMING"] if __import__("os").environ.get("MLN_POTRF_TIMING") else [])


# per-source extras: the one-wave-per-SIMD row-minimum sweep keeps its MFMA accumulators in architectural VGPRs (its vector
# epilogue reads them directly; in AGPRs every element costs a v_accvgpr_read)
EXTRA_FLAGS = {"rowmin_w64.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _deps(depfile, src):
    """Prerequisites recorded by -MMD (the source itself if there is no record yet)."""
    if not os.path.exists(depfile):
        return None
    text = open(depfile).read().replace("\\\n", " ")
    _, _, rhs = text.partition(":")
    deps = [d for d in rhs.split() if d]
    return deps or [src]


def _stale(target, deps):
    if not os.path.exists(target) or deps is None:
        return True
    t = os.path.getmtime(target)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        dep = obj[:-2] + ".d"
        if force or _stale(obj, _deps(dep, src)):
            jobs.append((src, obj, dep))

    def cc(job):
        src, obj, dep = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-MMD", "-MF", dep, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    with ThreadPoolExecutor(max_workers=min(max(8, (os.cpu_count() or 8) // 2), max(1, len(jobs)))) as ex:
        for (src, obj, dep), r in ex.map(cc, jobs):
            if verbose and (r.stdout or r.stderr):
                sys.stderr.write(r.stdout + r.stderr)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}")
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose and (r.stdout or r.stderr):
            sys.stderr.write(r.stdout + r.stderr)
        if r.returncode != 0:
            raise RuntimeError("link of libmellon_hip.so failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
