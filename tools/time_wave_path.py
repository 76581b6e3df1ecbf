"""Wall time of the one-wavefront-per-start maximiser (N <= 512) at a few shapes; SLS_HIP_LIB selects another build of
libsls_hip.so (A/B against an older library on the same box)."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem, synth_candidates
from oracle import oracle_py as oracle
m = sls()
ctx = m.Context(0)
for (N, D, S, nl) in ((128, 4, 10, 200), (61, 32, 1, 320), (100, 8, 64, 100), (300, 16, 1024, 50), (500, 32, 4096, 30), (128, 4, 10, 200)):
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    gp = m.GP(ctx, X, y, theta, b, 1)
    r = gp.acq_maximize(starts, nl)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = gp.acq_maximize(starts, nl)
        ts.append((time.perf_counter() - t0) * 1e3)
    dt = sum(ts) / 5
    st = gp.last_stats()
    print(f"N={N} D={D} S={S} n_local={nl}: {dt:.3f} ms per call, issued {st['evals_issued']} evaluations, best {r['value']:.9g}  (per call: {' '.join('%.2f' % t for t in ts)})")
    gp.close()
