"""Pin the oracle against the ONLY absolute golden numbers the reference holds
on this path (tests/test_reference_results.py:26-63,93-130, atol=1e-5)."""
import json
import os

import numpy as np
import pytest

from oracle import jax_prng as jp
from oracle import mellon_oracle as mo

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_results.json")))


def _inputs():
    # tests/test_reference_results.py:11-16 (threefry, partitionable bit layout)
    k1, k2, k3 = jp.split(jp.prng_key(42), 3, partitionable=True)
    X = jp.normal64(k1, (50, 2), True)
    y = jp.normal64(k2, (50, 3), True)
    Xt = jp.normal64(k3, (10, 2), True)
    return X, y, Xt


def test_threefry_known_answers():
    # Random123 kat_vectors, threefry2x32 20 rounds
    z = np.zeros(1, dtype=np.uint32)
    a, b = jp.threefry2x32((0, 0), z, z)
    assert (int(a[0]), int(b[0])) == (0x6B200159, 0x99BA4EFE)
    f = np.full(1, 0xFFFFFFFF, dtype=np.uint32)
    a, b = jp.threefry2x32((0xFFFFFFFF, 0xFFFFFFFF), f, f)
    assert (int(a[0]), int(b[0])) == (0x1CB996FC, 0xBB002BE7)
    a, b = jp.threefry2x32((0x13198A2E, 0x03707344),
                           np.array([0x243F6A88], dtype=np.uint32),
                           np.array([0x85A308D3], dtype=np.uint32))
    assert (int(a[0]), int(b[0])) == (0xC4923A9C, 0x483DF7A0)


def test_full_gp_reference_predictions():
    X, y, Xt = _inputs()
    pred = mo.function_fit(X, y, 1.0, n_landmarks=0)(Xt)
    assert np.allclose(pred, np.array(GOLD["full"]["expected_pred"]), atol=1e-5)
    # far tighter than the reference's own tolerance: 8 printed digits
    assert np.abs(pred - np.array(GOLD["full"]["expected_pred"])).max() < 5e-8


def test_full_gp_reference_leverage():
    # tests/test_leverage.py:26-44: leverage == diag(K (K + sigma^2 I)^-1)
    X, y, _ = _inputs()
    ls = mo.compute_ls(mo.exact_nn_distances(X))
    K = mo.Matern52(ls)(X, X)
    lev = np.diag(K @ np.linalg.inv(K + np.eye(50)))
    assert np.allclose(lev, np.array(GOLD["full"]["expected_lev"]), atol=1e-5)


def test_full_gp_reference_obs_variance_and_leverage_method():
    # tests/test_reference_results.py:21-63: est.predict.leverage(X), est.predict.obs_variance(X_test)
    X, y, Xt = _inputs()
    pred = mo.function_fit(X, y, 1.0, n_landmarks=0, obs_variance=True)
    assert np.allclose(pred.leverage(X), np.array(GOLD["full"]["expected_lev"]), atol=1e-5)
    assert np.allclose(pred.obs_variance(Xt), np.array(GOLD["full"]["expected_obsvar"]), atol=1e-5)


def test_sparse_gp_reference_leverage_and_obs_variance():
    # tests/test_reference_results.py:88-130 (15 k-means landmarks)
    X, y, Xt = _inputs()
    pred = mo.function_fit(X, y, 1.0, n_landmarks=15, obs_variance=True)
    assert np.allclose(pred.leverage(X), np.array(GOLD["sparse"]["expected_lev"]), atol=1e-5)
    assert np.allclose(pred.obs_variance(Xt), np.array(GOLD["sparse"]["expected_obsvar"]), atol=1e-5)


def test_sparse_gp_reference_predictions():
    X, y, Xt = _inputs()
    pred = mo.function_fit(X, y, 1.0, n_landmarks=15)(Xt)
    assert np.allclose(pred, np.array(GOLD["sparse"]["expected_pred"]), atol=1e-5)


def test_non_partitionable_layout_does_not_match():
    """Documents which threefry layout the golden numbers were made with."""
    k1, k2, k3 = jp.split(jp.prng_key(42), 3, partitionable=False)
    X = jp.normal64(k1, (50, 2), False)
    y = jp.normal64(k2, (50, 3), False)
    Xt = jp.normal64(k3, (10, 2), False)
    pred = mo.function_fit(X, y, 1.0, n_landmarks=0)(Xt)
    assert np.abs(pred - np.array(GOLD["full"]["expected_pred"])).max() > 1e-2
