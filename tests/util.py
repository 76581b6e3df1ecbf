"""Synthetic inputs of SURVEY.md 8(d) / BASELINE.md 3 (same seeded SplitMix64 generator as the oracle's C side)."""
import importlib

import numpy as np


def sls():
    return importlib.import_module("sequential-line-search_amd")


def synth_problem(oracle, D, N, seed=1234, lengthscale=None):
    X = oracle.fill_uniform(D * N, seed).reshape((D, N), order="F")
    noise = oracle.fill_normal(N, seed + 1)
    y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * noise
    ell = 0.5 * np.sqrt(max(D, 8) / 8.0) if lengthscale is None else lengthscale
    theta = np.concatenate([[0.5], np.full(D, ell)])
    return X, y, theta, 0.005


def synth_candidates(oracle, D, M, seed=1236):
    return oracle.fill_uniform(D * M, seed).reshape((D, M), order="F")


def relerr(a, b, floor=1e-300):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def assert_starts_agree(rg, ro, min_frac=0.95, margin_tol=1e-6, basin_rtol=1e-3):
    """Per-start end values of the HIP maximiser (rg) against the oracle run with diag=True (ro).

    Both sides run the same bounded L-BFGS statement by statement; they differ only in summation order.  A start can end
    elsewhere only where a DISCRETE decision of the algorithm sat within rounding of its threshold and the two sides took
    different branches: the Armijo test  ft <= f + c1 g.s  (measured on MI355X: 1 of 96 starts, margin 3.5e-9;
    tools/diverging_starts.py), or a clamp / active-bound / curvature test of a start that runs along the box boundary.
    Asserted: at least `min_frac` of the starts agree to 1e-6; a start that does not agree either has a near-threshold Armijo
    test on the oracle side (relative margin < margin_tol) or still ends in the same basin (within basin_rtol) -- the
    chosen maximiser itself is held to 1e-6 by the callers."""
    scale = max(np.abs(ro["y_stars"]).max(), 1e-300)
    agree = np.isclose(rg["y_stars"], ro["y_stars"], rtol=1e-6, atol=1e-12 * scale)
    assert agree.mean() >= min_frac, f"only {agree.mean():.2%} of the starts end at the oracle's value"
    for i in np.nonzero(~agree)[0]:
        near = np.isclose(rg["y_stars"][i], ro["y_stars"][i], rtol=basin_rtol, atol=1e-9 * scale)
        assert ro["armijo_margin"][i] < margin_tol or near, (
            f"start {i} ends at {rg['y_stars'][i]!r} vs oracle {ro['y_stars'][i]!r}: not the same basin, and no Armijo test was "
            f"closer than {ro['armijo_margin'][i]:.2e} to its threshold")
    return agree
