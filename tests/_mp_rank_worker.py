"""One rank of the multi-PROCESS test (tests/test_gpu_multiprocess.py): a sharded DensityEstimator fit under the
launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*), results to <out_dir>/rank<r>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_dir):
    import mellon_amd
    from mellon_amd import distributed
    from oracle import mellon_oracle as mo          # (test infrastructure: the shared synthetic inputs)
    comm = distributed.init_from_env()
    n, d, m = 24000, 8, 300
    x = mo.gaussian_mixture(n, d, seed=5)
    lo, hi = distributed.shard_bounds(n, comm.world_size, comm.rank)
    est = mellon_amd.DensityEstimator(n_landmarks=m, check_rank=False)      # landmarks and nn distances inside the fit
    dens = est.fit_predict(np.ascontiguousarray(x[lo:hi]))
    pred = est.predict(x[::41] + 0.01)
    np.savez(os.path.join(out_dir, f"rank{comm.rank}.npz"), dens=dens, pred=pred, landmarks=np.asarray(est.landmarks),
             ls=est.ls, mu=est.mu, lo=lo, hi=hi, backend=str(comm.backend),
             self_test_ok=bool(comm.self_test_report.get("ok")))
    comm.barrier()


if __name__ == "__main__":
    main(sys.argv[1])
