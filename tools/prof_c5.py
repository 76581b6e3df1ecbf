"""C5 (N=4096, D=128, Matern-5/2) MAP objective+gradient evaluations, for a rocprofv3 kernel trace."""
import importlib, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sls = importlib.import_module("sequential-line-search_amd")
N, D = int(os.environ.get("C5_N", 4096)), int(os.environ.get("C5_D", 128))
rng = np.random.default_rng(1234)
X = np.asfortranarray(rng.uniform(0, 1, (D, N)))
y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * rng.standard_normal(N)
ctx = sls.Context(0)
nll = sls.Nll(ctx, X, sls.KERNEL_MATERN52)
x = np.concatenate([[0.5, 0.005], np.full(D, 0.5 * np.sqrt(D / 8.0))])
nll.gp_objective(y, x)
ctx.synchronize()
t0 = time.perf_counter()
for i in range(10):
    v, g = nll.gp_objective(y, x * (1.0 + 0.01 * i))
ctx.synchronize()
print("C5 N=%d D=%d: %.3f ms per objective+gradient evaluation" % (N, D, (time.perf_counter() - t0) / 10 * 1e3))
