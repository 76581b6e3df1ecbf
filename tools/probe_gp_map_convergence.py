import sys, numpy as np, importlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
sls = importlib.import_module("sequential-line-search_amd")
ctx = sls.Context()
rng = np.random.default_rng(3)
for (D, N) in [(2, 40), (8, 100), (16, 128)]:
    X = rng.uniform(0, 1, (D, N)); y = np.sin(3 * X.sum(axis=0)) + 0.05 * rng.normal(size=N)
    n = D + 2
    z0 = np.log(np.concatenate([[0.5, 0.01], np.full(D, 0.5)]))
    lower = np.full(n, np.log(1e-8)); upper = np.full(n, np.log(10.0))
    h = sls.Nll(ctx, X, 1)
    vals = {}
    for k in [5, 10, 20, 30, 40, 60, 80, 120, 160, 240, 320, 500, 1000]:
        r = h.gp_map_fit(y, z0, lower, upper, k)
        vals[k] = (r["value"], r["evals"])
    h.close()
    ref = vals[1000][0]
    print(D, N, {k: ("%.3e" % ((ref - v[0]) / abs(ref)), v[1]) for k, v in vals.items()})
