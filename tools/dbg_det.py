import sys, numpy as np, hashlib
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem, synth_candidates
from oracle import oracle_py as oracle
def h(a): return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:8]
ctx = sls().Context(0)
for (D, N, S) in ((1, 5, 100), (4, 3, 10), (5, 120, 96), (2, 9, 32)):
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    out = []
    for rep in range(3):
        gp = sls().GP(ctx, X, y, theta, b, 1)
        r = gp.acq_maximize(starts, 20)
        v, g = gp.acq_eval(starts)
        nl = sls().Nll(ctx, X, 1)
        x = np.concatenate([[0.6, 0.01], np.full(D, 0.45)])
        ov, og = nl.gp_objective(y, x)
        pv, pg = nl.pref_objective([[0, 1, 2]] if N >= 3 else [[0]], y * 0.01)
        out.append((h(gp.matrix(1)), h(gp.matrix(3)), h(v), h(g), h(r["x_stars"]), h(r["y_stars"]), h(np.array([ov])), h(og), h(np.array([pv])), h(pg)))
        gp.close(); nl.close()
    print(D, N, S)
    for o in out: print("   ", o)
