import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem, synth_candidates
from oracle import oracle_py as oracle
m = sls(); ctx = m.Context(0)
X, y, theta, b = synth_problem(oracle, 8, 60)
gp = m.GP(ctx, X, y, theta, b, 1)
xs1 = synth_candidates(oracle, 8, 1); xs = synth_candidates(oracle, 8, 1000)
for flag in ("1", "0"):
    os.environ["SLS_WAVE_PATH"] = flag; m.tuning_reload()
    for name, f in (("predict M=1", lambda: gp.predict(xs1)), ("acq_eval+grad M=1", lambda: gp.acq_eval(xs1)), ("acq_eval+grad M=1000", lambda: gp.acq_eval(xs))):
        f(); t0 = time.perf_counter()
        for _ in range(200): f()
        print(f"SLS_WAVE_PATH={flag} {name}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call")
