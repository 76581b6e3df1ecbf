"""Stage timings for the BASELINE configs C2 (fit+predict) and C5 (MAP objective+gradient); wall clock with syncs."""
import sys, time, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem, synth_candidates
from oracle import oracle_py as oracle
m = sls()
ctx = m.Context(0)
def wall(f, reps=3):
    f(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
# C2
D, N, M = 16, 2048, 4096
X, y, theta, b = synth_problem(oracle, D, N)
Xs = synth_candidates(oracle, D, M)
gp = m.GP(ctx, X, y, theta, b, 0)
ctx.prof_enable(True); ctx.prof_reset()
t_create = wall(lambda: m.GP(ctx, X, y, theta, b, 0).close())
print("C2 create (incl. upload, alloc):", round(t_create, 3), "ms", {n: round(ctx.prof_get(n)[0] / 4, 3) for n in ("gram", "potrf", "trtri", "lauum")})
ctx.prof_reset()
t_pred = wall(lambda: gp.predict(Xs))
print("C2 predict 4096 (incl. PCIe):", round(t_pred, 3), "ms", {n: round(ctx.prof_get(n)[0] / 4, 3) for n in ("cross_gram", "var_gemm", "acq_gemm", "finalize")})
gp.close()
# fit at several N
for N in (1024, 4096, 8192):
    X, y, theta, b = synth_problem(oracle, 64, N)
    ctx.prof_reset()
    g = m.GP(ctx, X, y, theta, b, 1); g.close()
    print("fit N=%d D=64:" % N, {n: round(ctx.prof_get(n)[0], 3) for n in ("gram", "potrf", "trtri", "lauum")})
# large batched prediction: the triangular contraction (var_gemm) against its N^2 M algorithmic flops
D, N, M = 64, 8192, 32768
X, y, theta, b = synth_problem(oracle, D, N)
Xs = synth_candidates(oracle, D, M)
gp = m.GP(ctx, X, y, theta, b, 1)
gp.predict(Xs); ctx.prof_reset()
gp.predict(Xs)
t, n = ctx.prof_get("var_gemm")
print("predict N=8192 M=32768: var_gemm %.2f ms over %d launches = %.1f TFLOP/s on N^2 M flops (K^-1 form: 2 N^2 M)" % (t, n, N * N * M / (t * 1e-3) / 1e12))
gp.close()
# C5
D, N = 128, 4096
X, y, theta, b = synth_problem(oracle, D, N)
h = m.Nll(ctx, X, 1)
x = np.concatenate([[0.5, 0.005], np.full(D, theta[1])])
k = [0]
def ev():
    k[0] += 1
    xx = x.copy(); xx[2] *= (1 + 1e-3 * k[0])    # defeat the factorisation cache
    h.gp_objective(y, xx)
print("C5 MAP objective+gradient N=4096 D=128:", round(wall(ev), 3), "ms per evaluation")
