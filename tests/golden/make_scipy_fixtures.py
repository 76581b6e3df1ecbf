#!/usr/bin/env python3
"""Generate tests/golden/scipy_pipelines.npz -- complete GP pipelines at N in {130, 300, 512, 2048}.

The mpmath fixtures of make_fixtures.py stop at N = 10; every N <= 64 takes the UNBLOCKED branch of the oracle's
Cholesky, inverse and batched evaluation.  These cases pin the blocked branches (NB = 64 panels, 16-wide evaluation
blocks) and the padded 128-tile paths of the HIP library against an independent implementation:

  * kernels, their x-derivatives, EI / GP-UCB: numpy, written from the published definitions;
  * factorisation and solves: scipy.linalg.cho_factor / cho_solve / solve_triangular (LAPACK), with two steps of
    iterative refinement in numpy longdouble (80-bit) for alpha and for the candidate solves, so the stored answers are
    accurate to ~1e-13 relative even where kappa(K_y) ~ 1e5;
  * nothing here reads /root/reference, the oracle or the HIP library.

Stored per case: the inputs (X, y, theta, b, Xs, kernel) and alpha, log|K_y|, best_index, mu_best, diag(L), three
full rows of L, diag(K_y^-1), three full rows of K_y^-1, and mu / sigma / dmu / dsigma / EI / dEI / UCB / dUCB at M
candidates.  Run:  python tests/golden/make_scipy_fixtures.py
"""
import os

import numpy as np
from scipy.linalg import cho_factor, cho_solve, solve_triangular
from scipy.special import erfc

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scipy_pipelines.npz")
UCB_H = 2.0


def kernel_and_weight(kind, XA, XB, theta):
    """K[i, j] = k(xa_i, xb_j) and the derivative weight C with dk/dxa_d = -C * (xa_d - xb_d) / l_d^2.  XA: (D, A), XB: (D, B)."""
    a, ell = theta[0], theta[1:]
    d = (XA[:, :, None] - XB[:, None, :]) / ell[:, None, None]
    q = np.sum(d * d, axis=0)
    if kind == 0:
        K = a * np.exp(-0.5 * q)
        return K, K
    s = np.sqrt(5.0 * q)
    e = np.exp(-s)
    return a * (1.0 + s + 5.0 * q / 3.0) * e, a * (5.0 / 3.0) * (1.0 + s) * e


def refine(K, cf, rhs, steps=2):
    """x = K^-1 rhs with iterative refinement; residuals in longdouble."""
    x = cho_solve(cf, rhs)
    Kl = K.astype(np.longdouble)
    for _ in range(steps):
        r = (rhs.astype(np.longdouble) - Kl @ x.astype(np.longdouble)).astype(np.float64)
        x = x + cho_solve(cf, r)
    return x


def pipeline(kind, X, y, theta, b, Xs):
    D, N = X.shape
    M = Xs.shape[1]
    a, ell = theta[0], theta[1:]
    Kf, _ = kernel_and_weight(kind, X, X, theta)
    Kf = 0.5 * (Kf + Kf.T)
    np.fill_diagonal(Kf, a)
    K = Kf + b * np.eye(N)
    cf = cho_factor(K, lower=True)
    L = np.tril(cf[0])
    alpha = refine(K, cf, y)
    Kinv = cho_solve(cf, np.eye(N))
    Kinv = 0.5 * (Kinv + Kinv.T)
    mu_data = Kf @ alpha                      # PredictMu at every data point (regressor.cpp:29-43)
    best = int(np.argmax(mu_data))
    mu_best = float(mu_data[best])
    Ks, Cs = kernel_and_weight(kind, Xs, X, theta)          # (M, N)
    W = refine(K, cf, Ks.T.copy()).T                          # rows: K^-1 k for each candidate
    mu = Ks @ alpha
    s2 = a - np.sum(Ks * W, axis=1)
    sigma = np.sqrt(np.maximum(s2, 0.0))
    dmu = np.empty((D, M))
    dsg = np.empty((D, M))
    for d_ in range(D):
        diff = (Xs[d_][:, None] - X[d_][None, :]) / ell[d_] ** 2       # (M, N)
        J = -Cs * diff                                                   # dk_i/dx_d
        dmu[d_] = J @ alpha
        dsg[d_] = -np.sum(J * W, axis=1) / sigma
    u = (mu - mu_best) / sigma
    Phi = 0.5 * erfc(-u / np.sqrt(2.0))
    phi = np.exp(-0.5 * u * u) / np.sqrt(2.0 * np.pi)
    ei = (mu - mu_best) * Phi + sigma * phi
    dei = Phi * dmu + phi * dsg
    rows = np.array([N - 1, N // 2, min(129, N - 1)])
    return dict(alpha=alpha, logdet=2.0 * np.sum(np.log(np.diag(L))), best_index=best, mu_best=mu_best,
                L_diag=np.diag(L).copy(), L_rows=L[rows], Kinv_diag=np.diag(Kinv).copy(), Kinv_rows=Kinv[rows], rows=rows,
                mu=mu, sigma=sigma, dmu=dmu, dsigma=dsg, ei=ei, dei=dei, ucb=mu + UCB_H * sigma, ducb=dmu + UCB_H * dsg)


def main():
    rng = np.random.default_rng(20260930)
    out = {}
    cases = []
    for (D, N, M) in ((3, 130, 40), (6, 300, 48), (10, 512, 48), (8, 2048, 64)):
        X = rng.uniform(0, 1, (D, N))
        y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * rng.normal(size=N)
        theta = np.concatenate([[0.5], 0.5 * np.sqrt(max(D, 8) / 8.0) * rng.uniform(0.8, 1.25, D)])
        b = 0.005
        Xs = rng.uniform(0, 1, (D, M))
        Xs[:, 0] = X[:, N // 3]                      # one candidate exactly on a data point
        for kind in (0, 1):
            name = f"k{kind}_N{N}"
            cases.append(name)
            res = pipeline(kind, X, y, theta, b, Xs)
            res.update(X=X, y=y, theta=theta, b=b, Xs=Xs, kernel=kind)
            for k, v in res.items():
                out[f"{name}/{k}"] = np.asarray(v)
            print(name, "logdet", res["logdet"], "best", res["best_index"], "min sigma", res["sigma"].min())
    out["cases"] = np.array(cases)
    out["ucb_h"] = np.array(UCB_H)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
