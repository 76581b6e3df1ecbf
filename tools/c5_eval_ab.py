"""C5 evaluation (Matern-5/2 MAP objective + gradient, N = 4096, D = 128) timed alone: same-box A/B of library variants.
Prints ms per evaluation (mean of 3 x 20), the value and a digest of the gradient, and the relative difference to the oracle-free
reference run given by --ref (a .npy written by an earlier call with --save)."""
import argparse, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem
from oracle import oracle_py as oracle
ap = argparse.ArgumentParser(); ap.add_argument("--save"); ap.add_argument("--ref"); ap.add_argument("--N", type=int, default=4096); ap.add_argument("--D", type=int, default=128)
a = ap.parse_args()
m = sls(); ctx = m.Context(0)
D, N = a.D, a.N
X, y, theta, b = synth_problem(oracle, D, N)
h = m.Nll(ctx, X, 1)
x = np.concatenate([[0.5, 0.005], np.full(D, theta[1])])
k = [0]
def ev():
    k[0] += 1
    xx = x.copy(); xx[2] *= (1 + 1e-3 * k[0])
    return h.gp_objective(y, xx)
for _ in range(3): ev()
ts = []
for _ in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ev()
    ctx.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
v, g = h.gp_objective(y, x)
g = np.asarray(g)
print(f"N={N} D={D} ms_per_evaluation {min(ts):.4f} (runs {' '.join(f'{t:.4f}' for t in ts)})  value {v!r}  |grad| {np.linalg.norm(g)!r}", flush=True)
if a.save: np.save(a.save, np.concatenate([[v], g]))
if a.ref:
    r = np.load(a.ref)
    print(f"  vs {a.ref}: value rel {abs(v - r[0]) / abs(r[0]):.3e}  grad rel (max-norm) {np.max(np.abs(g - r[1:])) / np.max(np.abs(r[1:])):.3e}", flush=True)
h.close(); ctx.close()
