import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from oracle import oracle_py as orc
os.environ["OMP_NUM_THREADS"] = "64"
for Ns in (8192,):
    D = 64
    X = np.asfortranarray(np.random.default_rng(1).uniform(0, 1, (D, Ns)))
    y = np.exp(-np.sum((X - 0.4) ** 2, axis=0))
    theta = np.concatenate([[0.5], np.full(D, 0.5 * np.sqrt(8.0))])
    t0 = time.perf_counter(); ref = orc.Regressor(X, y, theta, 0.005, kernel=1); t1 = time.perf_counter()
    starts = np.asfortranarray(np.random.default_rng(2).uniform(0, 1, (D, 1024)))
    ref.acq_maximize(starts, 3, n_threads=64); t2 = time.perf_counter()
    print(Ns, "fit %.2f s" % (t1 - t0), "1024x3 evals %.2f s" % (t2 - t1))
