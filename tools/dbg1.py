import sys, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem, synth_candidates
from oracle import oracle_py as oracle
D, N, S, n_local = 5, 120, 96, 25
X, y, theta, b = synth_problem(oracle, D, N)
starts = synth_candidates(oracle, D, S)
ctx = sls().Context(0)
for kernel in (0, 1):
  for acq in (0, 1):
    ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    ro = ref.acq_maximize(starts, n_local, acq, 2.0)
    rg = gp.acq_maximize(starts, n_local, acq, 2.0)
    o = np.argsort(-ro["y_stars"])[:5]
    print("kernel", kernel, "acq", acq, "idx", rg["index"], ro["index"], "val", rg["value"], ro["value"])
    print("  oracle top", o, ro["y_stars"][o])
    print("  gpu    at ", o, rg["y_stars"][o])
    print("  x gpu", rg["x"], "\n  x ora", ro["x"])
    d = np.abs(rg["y_stars"] - ro["y_stars"]) / np.maximum(np.abs(ro["y_stars"]), 1e-300)
    print("  rel diff y_stars: max", d.max(), "n>1e-6:", (d > 1e-6).sum(), "n>1e-9:", (d > 1e-9).sum())
    dx = np.abs(rg["x_stars"] - ro["x_stars"]).max(axis=0)
    print("  max dx per start: max", dx.max(), "n>1e-6", (dx > 1e-6).sum())
    gp.close()
ctx.close()
