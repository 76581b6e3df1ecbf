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
