#!/usr/bin/env python3
"""Generate tests/golden/map_optima.npz -- independent optima of the two MAP fits of the path.

  * GP hyper-parameter MAP (reference: src/gaussian-process-regressor.cpp:18-24 priors, :36-193 objective + gradient,
    :274-299 driver = DIRECT(300) then TNEWTON(1000) on [1e-8, 50]^(D+2)), variables (a, b, r_1..r_D);
  * preference MAP (reference: src/preference-regressor.cpp:53-291 objective + gradients, :332-403 driver = TNEWTON from
    x_ini, bounds [-10, 10] for the goodness values and [1e-8, 10] for the hyper-parameters), with and without the joint
    estimation of the hyper-parameters.

The reference's drivers are NLopt's; the library replaces TNEWTON by a bounded L-BFGS (NLopt is not in the image), so an
iterate-by-iterate comparison is impossible.  What CAN be pinned is the optimum: here scipy's bounded truncated Newton
(minimize(method="TNC")) and L-BFGS-B maximise a numpy/scipy implementation of the same objectives -- kernels written from the
published definitions, LAPACK cho_factor / cho_solve, analytic gradients checked against finite
differences at generation time -- from the reference's initial point and from a set of seeded random starts; the best
optimum found and its objective value are stored, together with every distinct local optimum the starts reached (the GP
objective is multi-modal: DIRECT(300) in D + 2 dimensions decides the basin, and the library's DIRECT need not land in the basin
scipy's best start found).  Nothing here reads /root/reference, the oracle or the HIP library.

The functions below are also imported by tests/test_gpu_map_fit.py to evaluate the numpy objective at the point the
library returns (the library may legitimately end in a better local optimum than scipy: the test then checks stationarity
with THIS gradient instead of equality of the arguments).
Run:  python tests/golden/make_map_optima.py
"""
import os

import numpy as np
from scipy.linalg import cho_factor, cho_solve
from scipy.optimize import minimize

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "map_optima.npz")
LOG_LO, LOG_HI = np.log(1e-8), np.log(50.0)


def kernel_parts(kind, X, theta):
    """K_f, the derivative weight C ((dK/dl_p)_jk = C_jk (x_pj - x_pk)^2 / l_p^3) and dK/da, for X (D, N)."""
    a, ell = theta[0], theta[1:]
    d = (X[:, :, None] - X[:, None, :]) / ell[:, None, None]
    q = np.sum(d * d, axis=0)
    if kind == 0:
        E = np.exp(-0.5 * q)
        return a * E, a * E, E
    s = np.sqrt(5.0 * q)
    e = np.exp(-s)
    return a * (1.0 + s + 5.0 * q / 3.0) * e, a * (5.0 / 3.0) * (1.0 + s) * e, (1.0 + s + 5.0 * q / 3.0) * e


def log_lognormal(x, mu, s2):
    lx = np.log(x)
    return -lx - 0.5 * np.log(2.0 * np.pi * s2) - (lx - mu) ** 2 / (2.0 * s2)


def log_lognormal_d(x, mu, s2):
    return (mu - s2 - np.log(x)) / (s2 * x)


def gp_factor(kind, X, theta, b):
    Kf, C, Ea = kernel_parts(kind, X, theta)
    K = Kf + b * np.eye(X.shape[1])
    return K, C, Ea, cho_factor(K, lower=True)


def gp_map_objective(kind, X, y, x, want_grad=True):
    """log p(y | X, a, b, r) + log-normal priors at x = (a, b, r_1..r_D) (maximised); gradient wrt x."""
    D, N = X.shape
    a, b, r = x[0], x[1], x[2:]
    theta = np.concatenate([[a], r])
    try:
        K, C, Ea, cf = gp_factor(kind, X, theta, b)
    except np.linalg.LinAlgError:
        return -np.inf, np.zeros_like(x)
    alpha = cho_solve(cf, y)
    logdet = 2.0 * np.sum(np.log(np.diag(cf[0])))
    val = -0.5 * y @ alpha - 0.5 * logdet - 0.5 * N * np.log(2.0 * np.pi)
    val += log_lognormal(a, np.log(0.5), 0.5) + log_lognormal(b, np.log(1e-4), 0.5) + np.sum(log_lognormal(r, np.log(0.5), 0.5))
    if not want_grad:
        return val, None
    Kinv = cho_solve(cf, np.eye(N))
    W = np.outer(alpha, alpha) - Kinv
    g = np.empty_like(x)
    g[0] = 0.5 * np.sum(W * Ea) + log_lognormal_d(a, np.log(0.5), 0.5)
    g[1] = 0.5 * np.trace(W) + log_lognormal_d(b, np.log(1e-4), 0.5)
    G = 0.5 * W * C
    for p in range(D):
        dp = X[p][:, None] - X[p][None, :]
        g[2 + p] = np.sum(G * dp * dp) / r[p] ** 3 + log_lognormal_d(r[p], np.log(0.5), 0.5)
    return val, g


def pref_objective(kind, X, prefs, x, use_map, a0=0.5, r0=0.5, b0=0.005, prior_var=0.25, btl=0.01, want_grad=True):
    """Preference MAP objective at x = (y_1..y_M [, a, b, r_1..r_D]) (maximised); gradient wrt x."""
    D, M = X.shape
    y = x[:M]
    if use_map:
        a, b, r = x[M], x[M + 1], x[M + 2:]
    else:
        a, b, r = a0, b0, np.full(D, r0)
    theta = np.concatenate([[a], r])
    try:
        K, C, Ea, cf = gp_factor(kind, X, theta, b)
    except np.linalg.LinAlgError:
        return -np.inf, np.zeros_like(x)
    alpha = cho_solve(cf, y)
    logdet = 2.0 * np.sum(np.log(np.diag(cf[0])))
    val = -0.5 * y @ alpha - 0.5 * logdet - 0.5 * M * np.log(2.0 * np.pi)
    g = np.zeros_like(x)
    for p in prefs:
        f = y[list(p)] / btl
        m = np.max(f)
        lse = m + np.log(np.sum(np.exp(f - m)))
        val += f[0] - lse                                   # log BTL, evaluated stably (same value as the reference's form)
        if want_grad:
            w = np.exp(f - lse)
            for q, idx in enumerate(p):
                g[idx] += ((1.0 if q == 0 else 0.0) - w[q]) / btl
    if use_map:
        val += log_lognormal(a, np.log(a0), prior_var) + log_lognormal(b, np.log(b0), prior_var) + np.sum(log_lognormal(r, np.log(r0), prior_var))
    if not want_grad:
        return val, None
    g[:M] -= alpha
    if use_map:
        Kinv = cho_solve(cf, np.eye(M))
        W = np.outer(alpha, alpha) - Kinv
        g[M] = 0.5 * np.sum(W * Ea) + log_lognormal_d(a, np.log(a0), prior_var)
        g[M + 1] = 0.5 * np.trace(W) + log_lognormal_d(b, np.log(b0), prior_var)
        G = 0.5 * W * C
        for p in range(D):
            dp = X[p][:, None] - X[p][None, :]
            g[M + 2 + p] = np.sum(G * dp * dp) / r[p] ** 3 + log_lognormal_d(r[p], np.log(r0), prior_var)
    return val, g


def projected_grad_log(x, g, lo, hi, log_mask):
    """Gradient wrt the optimisation variables (log of the masked ones), components pushing out of the box zeroed."""
    gz = np.where(log_mask, g * x, g)
    z = np.where(log_mask, np.log(np.where(log_mask, x, 1.0)), x)
    out = gz.copy()
    out[(z <= lo + 1e-12) & (gz < 0)] = 0.0
    out[(z >= hi - 1e-12) & (gz > 0)] = 0.0
    return out


def _maximise(fun_z, z0s, bounds, found=None):
    """Best of TNC and L-BFGS-B over the starts; fun_z returns (value, grad) of the objective to MAXIMISE in z.
    found (optional list): every local optimum reached, as (value, z)."""
    best = None
    for z0 in z0s:
        for method, opts in (("TNC", dict(maxfun=20000, ftol=1e-15, gtol=1e-10, xtol=1e-15)),
                             ("L-BFGS-B", dict(maxiter=20000, maxfun=40000, ftol=1e-16, gtol=1e-10))):
            z = np.array(z0, float)
            for _ in range(3):                               # restarts: both stop early on flat stretches
                res = minimize(lambda zz: tuple(-v for v in fun_z(zz)), z, jac=True, method=method, bounds=bounds, options=opts)
                z = res.x
            v = fun_z(z)[0]
            if found is not None:
                found.append((v, z.copy()))
            if best is None or v > best[0]:
                best = (v, z.copy(), method)
    return best


def check_gradient(fun, x, h=1e-6):
    v, g = fun(x)
    for i in np.random.default_rng(0).choice(len(x), size=min(len(x), 6), replace=False):
        e = np.zeros_like(x)
        e[i] = h * max(1.0, abs(x[i]))
        fd = (fun(x + e)[0] - fun(x - e)[0]) / (2 * e[i])
        assert abs(fd - g[i]) <= 1e-5 * max(1.0, abs(fd)), (i, fd, g[i])


def synth(rng, D, N):
    X = rng.uniform(0, 1, (D, N))
    y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * rng.normal(size=N)
    return X, y


def make_pref_cases(rng, shapes, out, pref_cases):
    """Preference-MAP cases (D, M, number of tuples) appended to out / pref_cases, consuming rng in order."""
    for (D, M, npref) in shapes:
        X, f = synth(rng, D, M)
        prefs = []
        for _ in range(npref):
            idx = rng.choice(M, size=rng.integers(2, 4), replace=False)
            idx = idx[np.argsort(-f[idx])]                  # the best one first
            prefs.append([int(i) for i in idx])
        flat = np.array([i for p in prefs for i in p], dtype=np.uint32)
        offs = np.cumsum([0] + [len(p) for p in prefs]).astype(np.int32)
        for kind in (0, 1):
            for use_map in (0, 1):
                name = f"pref_k{kind}_M{M}_D{D}_map{use_map}"
                n = M + (2 + D if use_map else 0)
                log_mask = np.zeros(n, bool)
                log_mask[M:] = True
                x_chk = np.concatenate([0.3 * rng.normal(size=M), [0.4, 4e-3], rng.uniform(0.3, 0.8, D)])[:n]
                check_gradient(lambda x: pref_objective(kind, X, prefs, x, bool(use_map)), x_chk)

                def fz(z):
                    x = np.where(log_mask, np.exp(np.where(log_mask, z, 0.0)), z)
                    v, g = pref_objective(kind, X, prefs, x, bool(use_map))
                    return v, np.where(log_mask, g * x, g)
                z0 = np.zeros(n)
                if use_map:
                    z0[M:] = np.log(np.concatenate([[0.5, 0.005], np.full(D, 0.5)]))
                bounds = [(-10.0, 10.0)] * M + [(np.log(1e-8), np.log(10.0))] * (n - M)
                v, z, method = _maximise(fz, [z0], bounds)          # the reference's x_ini only (one basin; map0 is concave)
                x = np.where(log_mask, np.exp(np.where(log_mask, z, 0.0)), z)
                lo = np.array([bb[0] for bb in bounds]); hi = np.array([bb[1] for bb in bounds])
                pg = projected_grad_log(x, pref_objective(kind, X, prefs, x, bool(use_map))[1], lo, hi, log_mask)
                pref_cases.append(name)
                out[f"{name}/X"], out[f"{name}/prefs_flat"], out[f"{name}/offsets"] = X, flat, offs
                out[f"{name}/kernel"], out[f"{name}/use_map"] = np.array(kind), np.array(use_map)
                out[f"{name}/x_opt"], out[f"{name}/value"], out[f"{name}/pg_inf"] = x, np.array(v), np.array(np.max(np.abs(pg)))
                print(name, "value %.9f" % v, method, "pg_inf %.2e" % np.max(np.abs(pg)), "hyp", x[M:M + 4] if use_map else "-")


def main():
    rng = np.random.default_rng(20260929)
    out, gp_cases, pref_cases = {}, [], []
    for N in (20, 90, 300):
        for D in (1, 8, 32):
            X, y = synth(rng, D, N)
            for kind in (0, 1):
                name = f"gp_k{kind}_N{N}_D{D}"
                fx = lambda x: gp_map_objective(kind, X, y, x)
                check_gradient(fx, np.concatenate([[0.4, 3e-3], rng.uniform(0.3, 0.9, D)]))

                def fz(z):
                    x = np.exp(z)
                    v, g = gp_map_objective(kind, X, y, x)
                    return v, g * x
                starts = [np.log(np.concatenate([[0.5, 1e-4], np.full(D, 0.5)]))]           # the reference's x_ini (prior medians)
                # the modes of the three log-normal priors, and the same with a small signal variance: the "no dimension
                # matters" optimum every problem of this family has (all r on the prior's mode, the data explained by noise)
                mode = lambda med, var: np.log(med) - var
                starts.append(np.concatenate([[mode(0.5, 0.5), mode(1e-4, 0.5)], np.full(D, mode(0.5, 0.5))]))
                starts.append(np.concatenate([[np.log(5e-3), mode(1e-4, 0.5)], np.full(D, mode(0.5, 0.5))]))
                for _ in range(6):
                    starts.append(np.concatenate([rng.uniform(np.log(0.05), np.log(5.0), 1), rng.uniform(np.log(1e-6), np.log(1e-1), 1),
                                                  rng.uniform(np.log(0.1), np.log(10.0), D)]))
                found = []
                v, z, method = _maximise(fz, starts, [(LOG_LO, LOG_HI)] * (D + 2), found)
                x = np.exp(z)
                # distinct local optima (value to 1e-7 relative), best first
                found.sort(key=lambda t: -t[0])
                uniq = []
                for (fv, fzv) in found:
                    if np.isfinite(fv) and not any(abs(fv - u[0]) <= 1e-7 * max(1.0, abs(fv)) for u in uniq):
                        uniq.append((fv, fzv))
                out[f"{name}/local_values"] = np.array([u[0] for u in uniq])
                out[f"{name}/local_x"] = np.exp(np.array([u[1] for u in uniq]))
                pg = projected_grad_log(x, gp_map_objective(kind, X, y, x)[1], LOG_LO, LOG_HI, np.ones(D + 2, bool))
                gp_cases.append(name)
                out[f"{name}/X"], out[f"{name}/y"], out[f"{name}/kernel"] = X, y, np.array(kind)
                out[f"{name}/x_opt"], out[f"{name}/value"], out[f"{name}/pg_inf"] = x, np.array(v), np.array(np.max(np.abs(pg)))
                print(name, "value %.9f" % v, method, "a %.4g b %.4g r[:3]" % (x[0], x[1]), x[2:5], "pg_inf %.2e" % np.max(np.abs(pg)),
                      "local optima:", np.round(out[f"{name}/local_values"], 4))
    make_pref_cases(rng, ((2, 25, 12), (6, 60, 30)), out, pref_cases)
    out["gp_cases"], out["pref_cases"] = np.array(gp_cases), np.array(pref_cases)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


def main_c3():
    """Round 5: preference-MAP cases at BASELINE config 3's own shape (D = 32, M = 91 after 30 iterations), with and without the
    joint hyper-parameter estimation (the reference's default: use_MAP_hyperparams = true,
    include/sequential-line-search/sequential-line-search.hpp:37).  A file and a seed of their own, so that map_optima.npz stays
    byte-identical.  Run:  python tests/golden/make_map_optima.py c3"""
    rng = np.random.default_rng(20260930)
    out, pref_cases = {}, []
    make_pref_cases(rng, ((32, 91, 45), (32, 40, 20)), out, pref_cases)
    out["pref_cases"] = np.array(pref_cases)
    path = os.path.join(os.path.dirname(OUT), "map_optima_c3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    import sys
    main_c3() if sys.argv[1:] == ["c3"] else main()
