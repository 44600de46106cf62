"""Golden fixtures generated HERE by the oracle (NOT by the reference, which cannot be imported:
SURVEY.md S8c) -- they freeze the oracle's converged answers so that GPU parity runs on the GPU
box compare against committed numbers, and any later change of the oracle shows up as a diff.

  c1_density.npz      BASELINE config 1: 1000 x 10 mixture, all defaults -> full GP, Matern52
  sparse_density.npz  5000 x 10, 256 k-means landmarks, sparse_cholesky
  time_density.npz    1600 x 3 + time (4 time points), product Matern52, 64 landmarks, ls_time=1.5
Inputs are regenerated from seeds with oracle.mellon_oracle.gaussian_mixture; landmarks and
nn_distances (shared inputs: sklearn k-means / exact 1-NN) are stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mellon_oracle as mo  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, fit, **extra):
    np.savez_compressed(
        os.path.join(OUT, name), log_density_x=fit.log_density_x, pre_transformation=fit.pre_transformation,
        nn_distances=fit.nn_distances, mu=fit.mu, ls=fit.ls, d=fit.d, loss=fit.loss,
        landmarks=np.zeros((0, 0)) if fit.landmarks is None else fit.landmarks, **extra)
    print(name, "loss", fit.loss, "evals", fit.n_eval)


def main():
    x = mo.gaussian_mixture(1000, 10, seed=1)
    save("c1_density.npz", mo.density_fit(x, lbfgsb_options=mo.LBFGSB_TIGHT), seed=1, n=1000, dims=10)

    x = mo.gaussian_mixture(5000, 10, seed=12)
    fit = mo.density_fit(x, n_landmarks=256, lbfgsb_options=mo.LBFGSB_TIGHT)
    xq = mo.gaussian_mixture(300, 10, seed=13)
    save("sparse_density.npz", fit, seed=12, n=5000, dims=10, query_seed=13, predict_query=fit.predict(xq))

    rng = np.random.default_rng(4)
    xs = mo.gaussian_mixture(1600, 3, seed=4)
    times = np.repeat(np.arange(4.0), 400)
    xs = xs + 0.3 * times[:, None]
    xt = np.concatenate([xs, times[:, None]], axis=1)
    nn = mo.per_time_nn_distances(xs, times)
    ls = mo.compute_ls(nn)
    km = np.array(xt)
    km[:, -1] *= ls / 1.5
    lm = mo.compute_landmarks(km, mo.SPARSE_CHOLESKY, 64, 42)
    lm[:, -1] /= ls / 1.5
    fit = mo.density_fit(xt, n_landmarks=64, landmarks=lm, nn_distances=nn, d=3, ls=ls, ls_time=1.5,
                         lbfgsb_options=mo.LBFGSB_TIGHT)
    q = xt[rng.choice(1600, 200, replace=False)]
    save("time_density.npz", fit, seed=4, n=1600, dims=3, ls_time=1.5, x_time=xt, query=q, predict_query=fit.predict(q))


if __name__ == "__main__":
    main()
