"""C2 fit (N = 2048, D = 16, ARD-SE: upload + Gram + factor + inverse + alpha + summary) timed alone: ms per GP construction (wall)."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem
from oracle import oracle_py as oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
m = sls(); ctx = m.Context(0)
X, y, theta, b = synth_problem(oracle, D, N)
for _ in range(3): m.GP(ctx, X, y, theta, b, 0).close()
ts = []
for _ in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.GP(ctx, X, y, theta, b, 0).close()
    ctx.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
g = m.GP(ctx, X, y, theta, b, 0)
mu, sd = g.predict(X[:, :64].copy())
print(f"N={N} D={D} fit_ms_wall {min(ts):.4f} (runs {' '.join(f'{t:.4f}' for t in ts)})  digest mu {float(np.sum(mu))!r} sd {float(np.sum(sd))!r}", flush=True)
g.close(); ctx.close()
