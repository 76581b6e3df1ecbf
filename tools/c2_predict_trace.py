"""C2 predict (N = 2048, D = 16, 4096 query points): wall ms per call; under rocprofv3 the kernel timeline of one call."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem, synth_candidates
from oracle import oracle_py as oracle
m = sls(); ctx = m.Context(0)
D, N, M = 16, 2048, 4096
X, y, theta, b = synth_problem(oracle, D, N); Xs = synth_candidates(oracle, D, M)
gp = m.GP(ctx, X, y, theta, b, 0)
for _ in range(3): gp.predict(Xs)
ts = []
for _ in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(20): mu, sd = gp.predict(Xs)
    ctx.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
print(f"predict_ms_wall {min(ts):.4f} (runs {' '.join(f'{t:.4f}' for t in ts)}) digest {float(np.sum(mu))!r} {float(np.sum(sd))!r}", flush=True)
gp.close(); ctx.close()
