"""Wall time of the device-resident preference MAP fit (sls_pref_map_fit) at the C3 shapes, one launch vs one launch per
evaluation vs the host-driven optimiser (SLS_MAP_DEVICE=0 objective calls)."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls
m = sls(); ctx = m.Context(0)
for (M, D) in ((31, 32), (61, 32), (91, 32), (61, 8)):
    rng = np.random.default_rng(M)
    X = rng.uniform(0, 1, (D, M))
    prefs = [[3 * i + 1, 3 * i, 3 * i + 2] for i in range((M - 1) // 3)]
    z0, lo, hi = np.zeros(M), np.full(M, -10.0), np.full(M, 10.0)
    h = m.Nll(ctx, X, 1)
    for epl, label in ((0, "one launch"), (1, "launch per evaluation")):
        r = h.pref_map_fit(prefs, z0, lo, hi, 100, epl, r=0.5, a=0.5, b=0.001, btl_scale=0.01)
        t0 = time.perf_counter()
        for _ in range(20):
            r = h.pref_map_fit(prefs, z0, lo, hi, 100, epl, r=0.5, a=0.5, b=0.001, btl_scale=0.01)
        dt = (time.perf_counter() - t0) / 20 * 1e3
        print(f"M={M} D={D} {label}: {dt:.3f} ms per fit, {r['evals']} evaluations, {dt / r['evals'] * 1e3:.1f} us per evaluation, value {r['value']:.6f}")
    t0 = time.perf_counter()
    for _ in range(20):
        v = h.pref_objective(prefs, z0, r=0.5, a=0.5, b=0.001, btl_scale=0.01)
    print(f"M={M} D={D} single objective call: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us")
    h.close()
