"""One-off costs of a process that uses the library: library load, context creation (= HIP runtime initialisation: the floor of any
HIP process, tools/probes/hip_startup.hip), first and second call of the small-problem entry points.  Measured on MI355X:
load 12-27 ms | Context 150-230 ms | Nll create 5.5 | first objective 0.4, second 0.06 | first maximise 18, second 0.44 ms."""
import importlib, os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
t0 = time.perf_counter()
sls = importlib.import_module("sequential-line-search_amd")
sls.lib()
t1 = time.perf_counter()
ctx = sls.Context(0)
t2 = time.perf_counter()
rng = np.random.default_rng(0)
X = np.asfortranarray(rng.uniform(0, 1, (4, 20))); y = rng.normal(size=20)
h = sls.Nll(ctx, X, 1)
t3 = time.perf_counter()
x = np.array([0.5, 0.01, 0.5, 0.5, 0.5, 0.5])
h.gp_objective(y, x)
t4 = time.perf_counter()
h.gp_objective(y, x * 1.01)
t5 = time.perf_counter()
gp = sls.GP(ctx, X, y, np.array([0.5, .5, .5, .5, .5]), 0.01, 1)
t6 = time.perf_counter()
gp.acq_maximize(rng.uniform(0, 1, (4, 1)), 20)
t7 = time.perf_counter()
gp.acq_maximize(rng.uniform(0, 1, (4, 1)), 20)
t8 = time.perf_counter()
print("load lib %.1f ms | Context %.1f | Nll create %.1f | first objective %.1f | second %.2f | GP create (fit) %.1f | first maximise %.1f | second %.2f" %
      tuple(1e3 * v for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7)))
