"""Launch-bound regime: multi-start maximisation at SLS-demo sizes with and without hipGraph replay."""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem, synth_candidates
from oracle import oracle_py as oracle
m = sls(); ctx = m.Context(0)
for (D, N, S, nl) in ((32, 90, 10, 320), (8, 30, 10, 80), (1, 20, 100, 50)):
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    gp = m.GP(ctx, X, y, theta, b, 1)
    res = {}
    for flag in os.environ.get("SLS_TIME_FLAGS", "0,1").split(","):
        os.environ["SLS_USE_GRAPH"] = flag
        gp.acq_maximize(starts, nl)
        t0 = time.perf_counter()
        for _ in range(3): r = gp.acq_maximize(starts, nl)
        res[flag] = ((time.perf_counter() - t0) / 3 * 1e3, r)
    same = all(np.array_equal(res[k][1]["x_stars"], res["0"][1]["x_stars"]) for k in res)
    print(f"D={D} N={N} S={S} n_local={nl}: " + "  ".join(f"{k}:{v[0]:.2f}ms" for k, v in res.items()) + f"  identical={same}")
    gp.close()
