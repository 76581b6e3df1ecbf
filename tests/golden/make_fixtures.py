#!/usr/bin/env python3
"""Generate tests/golden/fixtures.json -- independent known answers that pin the oracle.

The reference cannot be built or imported in this image and holds no
golden vectors, so these fixtures come from INDEPENDENT implementations of the same
published definitions:
  * mpmath at 50 digits for kernel / EI / log-normal / BTL scalars and for complete
    small GP pipelines (posterior, gradients, marginal likelihood, preference objective);
  * scikit-learn's GaussianProcessRegressor (ConstantKernel*RBF / Matern(nu=2.5),
    alpha=b, optimizer=None) as a third opinion on mu / sigma.
Nothing here reads /root/reference or the oracle.  Run:  python tests/golden/make_fixtures.py
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures.json")


def F(x):
    return float(x)


def k_mp(kind, xa, xb, theta):
    a = mp.mpf(theta[0])
    q = sum(((mp.mpf(xa[i]) - mp.mpf(xb[i])) / mp.mpf(theta[1 + i])) ** 2 for i in range(len(xa)))
    if kind == 0:
        return a * mp.e ** (-q / 2)
    s = mp.sqrt(5 * q)
    return a * (1 + s + s * s / 3) * mp.e ** (-s)


def k_theta_deriv_mp(kind, xa, xb, theta):
    th = [mp.mpf(t) for t in theta]
    out = []
    for p in range(len(th)):
        out.append(mp.diff(lambda t: k_mp(kind, xa, xb, th[:p] + [t] + th[p + 1:]), th[p]))
    return out


def k_x_deriv_mp(kind, xa, xb, theta):
    xs = [mp.mpf(v) for v in xa]
    out = []
    for p in range(len(xs)):
        out.append(mp.diff(lambda t: k_mp(kind, xs[:p] + [t] + xs[p + 1:], xb, theta), xs[p]))
    return out


def Phi(u):
    return mp.ncdf(u)


def phi(u):
    return mp.npdf(u)


def gp_pipeline_mp(kind, X, y, theta, b, Xs):
    """Everything the hot path computes, at 50 digits.  X: (D,N) list-of-columns layout as numpy."""
    D, N = X.shape
    cols = [[mp.mpf(X[d, i]) for d in range(D)] for i in range(N)]
    K = mp.matrix(N, N)
    for i in range(N):
        for j in range(N):
            K[i, j] = k_mp(kind, cols[i], cols[j], theta) + (mp.mpf(b) if i == j else 0)
    Kinv = K ** -1
    yv = mp.matrix([mp.mpf(v) for v in y])
    alpha = Kinv * yv
    mu_data = [sum(k_mp(kind, cols[i], cols[j], theta) * alpha[j] for j in range(N)) for i in range(N)]
    best = max(range(N), key=lambda i: mu_data[i])
    mu_best = mu_data[best]
    a = mp.mpf(theta[0])
    res = dict(mu=[], sigma=[], dmu=[], dsigma=[], ei=[], dei=[], ucb=[], ducb=[])

    def mu_f(x):
        kv = mp.matrix([k_mp(kind, x, cols[j], theta) for j in range(N)])
        return (kv.T * alpha)[0]

    def sg_f(x):
        kv = mp.matrix([k_mp(kind, x, cols[j], theta) for j in range(N)])
        return mp.sqrt(a - (kv.T * Kinv * kv)[0])

    for m in range(Xs.shape[1]):
        x = [mp.mpf(Xs[d, m]) for d in range(D)]
        mu, sg = mu_f(x), sg_f(x)
        dmu = [mp.diff(lambda t: mu_f(x[:p] + [t] + x[p + 1:]), x[p]) for p in range(D)]
        dsg = [mp.diff(lambda t: sg_f(x[:p] + [t] + x[p + 1:]), x[p]) for p in range(D)]
        u = (mu - mu_best) / sg
        ei = (mu - mu_best) * Phi(u) + sg * phi(u)
        dei = [Phi(u) * dmu[p] + phi(u) * dsg[p] for p in range(D)]
        res["mu"].append(F(mu)); res["sigma"].append(F(sg))
        res["dmu"].append([F(v) for v in dmu]); res["dsigma"].append([F(v) for v in dsg])
        res["ei"].append(F(ei)); res["dei"].append([F(v) for v in dei])
        res["ucb"].append(F(mu + 2 * sg)); res["ducb"].append([F(dmu[p] + 2 * dsg[p]) for p in range(D)])
    res["alpha"] = [F(v) for v in alpha]
    res["best_index"] = int(best)
    res["mu_best"] = F(mu_best)
    res["logdet"] = F(mp.log(mp.det(K)))
    res["K"] = [[F(K[i, j]) for j in range(N)] for i in range(N)]
    res["Kinv"] = [[F(Kinv[i, j]) for j in range(N)] for i in range(N)]
    return res


def lognormal_mp(x, m, v):
    x, m, v = mp.mpf(x), mp.mpf(m), mp.mpf(v)
    return -mp.log(x) - mp.log(2 * mp.pi * v) / 2 - (mp.log(x) - m) ** 2 / (2 * v)


def gp_map_objective_mp(kind, X, y, xvec):
    """src/gaussian-process-regressor.cpp:141-193 definition, 50 digits, gradient by mp.diff."""
    D, N = X.shape
    cols = [[mp.mpf(X[d, i]) for d in range(D)] for i in range(N)]
    yv = mp.matrix([mp.mpf(v) for v in y])

    def obj(xv):
        a, b, r = xv[0], xv[1], xv[2:]
        theta = [a] + list(r)
        K = mp.matrix(N, N)
        for i in range(N):
            for j in range(N):
                K[i, j] = k_mp(kind, cols[i], cols[j], theta) + (b if i == j else 0)
        t1 = -(yv.T * (K ** -1) * yv)[0] / 2
        t2 = -mp.log(mp.det(K)) / 2
        t3 = -mp.mpf(N) / 2 * mp.log(2 * mp.pi)
        pri = lognormal_mp(a, mp.log(mp.mpf("0.5")), "0.5") + lognormal_mp(b, mp.log(mp.mpf("1e-4")), "0.5")
        for ri in r:
            pri += lognormal_mp(ri, mp.log(mp.mpf("0.5")), "0.5")
        return t1 + t2 + t3 + pri

    xv = [mp.mpf(v) for v in xvec]
    val = obj(xv)
    grad = [mp.diff(lambda t: obj(xv[:p] + [t] + xv[p + 1:]), xv[p]) for p in range(len(xv))]
    return F(val), [F(g) for g in grad]


def pref_objective_mp(kind, X, prefs, xvec, use_map, a0, r0, b0, pvar, s):
    """src/preference-regressor.cpp:129-259 definition, 50 digits."""
    D, M = X.shape
    cols = [[mp.mpf(X[d, i]) for d in range(D)] for i in range(M)]
    s = mp.mpf(s)

    def obj(xv):
        y = xv[:M]
        if use_map:
            a, b, r = xv[M], xv[M + 1], xv[M + 2:]
        else:
            a, b, r = mp.mpf(a0), mp.mpf(b0), [mp.mpf(r0)] * D
        theta = [a] + list(r)
        o = mp.mpf(0)
        for p in prefs:
            e = [mp.e ** (y[i] / s) for i in p]
            o += mp.log(e[0] / sum(e))
        K = mp.matrix(M, M)
        for i in range(M):
            for j in range(M):
                K[i, j] = k_mp(kind, cols[i], cols[j], theta) + (b if i == j else 0)
        yv = mp.matrix(y)
        o += -(yv.T * (K ** -1) * yv)[0] / 2 - mp.log(mp.det(K)) / 2 - mp.mpf(M) / 2 * mp.log(2 * mp.pi)
        if use_map:
            o += lognormal_mp(a, mp.log(mp.mpf(a0)), pvar) + lognormal_mp(b, mp.log(mp.mpf(b0)), pvar)
            for ri in r:
                o += lognormal_mp(ri, mp.log(mp.mpf(r0)), pvar)
        return o

    xv = [mp.mpf(v) for v in xvec]
    val = obj(xv)
    grad = [mp.diff(lambda t: obj(xv[:p] + [t] + xv[p + 1:]), xv[p]) for p in range(len(xv))]
    return F(val), [F(g) for g in grad]


def main():
    rng = np.random.default_rng(20260929)
    fx = {}

    # 1. kernel scalars
    ks = []
    for kind in (0, 1):
        for D in (1, 2, 5, 8):
            for _ in range(6):
                xa, xb = rng.uniform(0, 1, D), rng.uniform(0, 1, D)
                theta = np.concatenate([[rng.uniform(0.1, 2.0)], rng.uniform(0.05, 1.5, D)])
                ks.append(dict(kernel=kind, xa=xa.tolist(), xb=xb.tolist(), theta=theta.tolist(),
                               k=F(k_mp(kind, xa, xb, theta)),
                               dtheta=[F(v) for v in k_theta_deriv_mp(kind, list(xa), list(xb), list(theta))],
                               dx=[F(v) for v in k_x_deriv_mp(kind, list(xa), list(xb), list(theta))]))
        # coincident points: k = a, dk/dx = 0, dk/dl = 0 (Matern must be finite at s = 0)
        xa = rng.uniform(0, 1, 3)
        theta = np.array([0.7, 0.3, 0.4, 0.5])
        ks.append(dict(kernel=kind, xa=xa.tolist(), xb=xa.tolist(), theta=theta.tolist(), k=0.7,
                       dtheta=[1.0, 0.0, 0.0, 0.0], dx=[0.0, 0.0, 0.0]))
    fx["kernel_scalars"] = ks

    # 2. EI / normal pdf / cdf scalars
    eis = []
    for _ in range(40):
        mu, mb = rng.normal(0, 1), rng.normal(0, 1)
        sg = float(10 ** rng.uniform(-6, 0.5))
        u = (mp.mpf(mu) - mp.mpf(mb)) / mp.mpf(sg)
        eis.append(dict(mu=mu, sigma=sg, mu_best=mb, cdf=F(Phi(u)), pdf=F(phi(u)),
                        ei=F((mp.mpf(mu) - mp.mpf(mb)) * Phi(u) + mp.mpf(sg) * phi(u))))
    fx["ei_scalars"] = eis

    # 3. log-normal prior
    lns = []
    for _ in range(20):
        x = float(10 ** rng.uniform(-6, 1.5)); m = float(np.log(10 ** rng.uniform(-4, 0))); v = float(rng.uniform(0.05, 1.0))
        lns.append(dict(x=x, mu=m, sigma2=v, logpdf=F(lognormal_mp(x, m, v)),
                        dlogpdf=F(mp.diff(lambda t: lognormal_mp(t, m, v), mp.mpf(x)))))
    fx["lognormal"] = lns

    # 4. BTL
    btls = []
    for n in (2, 3, 5):
        for _ in range(5):
            f = rng.normal(0, 0.02, n); s = 0.01
            fm = [mp.mpf(v) for v in f]
            def btl(fv):
                e = [mp.e ** (t / mp.mpf(s)) for t in fv]
                return e[0] / sum(e)
            d = [mp.diff(lambda t: btl(fm[:p] + [t] + fm[p + 1:]), fm[p]) for p in range(n)]
            btls.append(dict(f=f.tolist(), scale=s, btl=F(btl(fm)), dbtl=[F(v) for v in d]))
    fx["btl"] = btls

    # 5. small GP pipelines at 50 digits
    pipes = []
    for kind in (0, 1):
        for (D, N, M) in ((1, 6, 5), (3, 10, 6)):
            X = rng.uniform(0, 1, (D, N))
            y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * rng.normal(size=N)
            theta = np.concatenate([[0.5], np.full(D, 0.5) * rng.uniform(0.8, 1.2, D)])
            b = 0.005
            Xs = rng.uniform(0, 1, (D, M))
            res = gp_pipeline_mp(kind, X, y, list(theta), b, Xs)
            res.update(kernel=kind, X=X.tolist(), y=y.tolist(), theta=theta.tolist(), b=b, Xs=Xs.tolist())
            pipes.append(res)
    fx["gp_pipelines"] = pipes

    # 6. sklearn third opinion (mu, latent std) at moderate N
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern
    sk = []
    for kind in (0, 1):
        D, N, M = 4, 60, 25
        X = rng.uniform(0, 1, (D, N))
        y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * rng.normal(size=N)
        ls = rng.uniform(0.3, 0.8, D); a = 0.5; b = 0.005
        kern = ConstantKernel(a, "fixed") * (RBF(ls, "fixed") if kind == 0 else Matern(ls, "fixed", nu=2.5))
        gpr = GaussianProcessRegressor(kernel=kern, alpha=b, optimizer=None).fit(X.T, y)
        Xs = rng.uniform(0, 1, (D, M))
        mu, sd = gpr.predict(Xs.T, return_std=True)
        sk.append(dict(kernel=kind, X=X.tolist(), y=y.tolist(), theta=[a] + ls.tolist(), b=b, Xs=Xs.tolist(),
                       mu=mu.tolist(), sigma=sd.tolist()))
    fx["sklearn_gp"] = sk

    # 7. GP MAP objective + gradient
    maps = []
    for kind in (0, 1):
        D, N = 2, 7
        X = rng.uniform(0, 1, (D, N))
        y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * rng.normal(size=N)
        xv = np.array([0.6, 0.01, 0.45, 0.7])
        val, grad = gp_map_objective_mp(kind, X, y, xv)
        maps.append(dict(kernel=kind, X=X.tolist(), y=y.tolist(), x=xv.tolist(), value=val, grad=grad))
    fx["gp_map"] = maps

    # 8. preference objective + gradient
    prefs_fx = []
    for kind in (0, 1):
        for use_map in (False, True):
            D, M = 2, 6
            X = rng.uniform(0, 1, (D, M))
            prefs = [[0, 1, 2], [3, 0, 4], [5, 3]]
            yv = rng.normal(0, 0.02, M)
            xv = np.concatenate([yv, [0.55, 0.006, 0.45, 0.6]]) if use_map else yv
            val, grad = pref_objective_mp(kind, X, prefs, xv, use_map, 0.5, 0.5, 0.005, 0.25, 0.01)
            prefs_fx.append(dict(kernel=kind, use_map=use_map, X=X.tolist(), prefs=prefs, x=xv.tolist(), value=val, grad=grad))
    fx["pref_objective"] = prefs_fx

    with open(OUT, "w") as f:
        json.dump(fx, f, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
