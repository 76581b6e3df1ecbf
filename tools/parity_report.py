"""Max relative errors of the HIP path vs the CPU oracle on the synthetic inputs of SURVEY.md 8(d) (target <= 1e-6)."""
import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem, synth_candidates
from oracle import oracle_py as oracle
m = sls(); ctx = m.Context(0)
def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
print("| config | kernel | mu | sigma | grad mu | grad sigma | EI | grad EI | UCB | maximiser x | maximiser value |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for (name, D, N, M, S, nl) in (("C1-like", 1, 20, 512, 64, 30), ("C3-like", 32, 90, 512, 64, 30), ("C2", 16, 2048, 4096, 256, 12),
                                ("N=1024 D=64", 64, 1024, 1024, 256, 12)):
    for kernel, kname in ((0, "ARD-SE"), (1, "Matern-5/2")):
        X, y, theta, b = synth_problem(oracle, D, N)
        Xs = synth_candidates(oracle, D, M)
        gp = m.GP(ctx, X, y, theta, b, kernel); ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
        mu, sg = gp.predict(Xs); muo, sgo = ref.predict_batch(Xs)
        dm, ds = gp.predict_grad(Xs); dmo, dso = ref.predict_grad_batch(Xs)
        ei, dei = gp.acq_eval(Xs, 0); eio, deio = ref.acq_eval_batch(Xs, 0)
        ucb = gp.acq_eval(Xs, 1, 2.0, want_grad=False); ucbo = ref.acq_eval_batch(Xs, 1, 2.0, want_grad=False)
        starts = Xs[:, :S]
        rg = gp.acq_maximize(starts, nl); ro = ref.acq_maximize(starts, nl)
        print(f"| {name} (N={N}, D={D}) | {kname} | {rel(mu, muo):.1e} | {rel(sg, sgo):.1e} | {rel(dm, dmo):.1e} | {rel(ds, dso):.1e} | "
              f"{rel(ei, eio):.1e} | {rel(dei, deio):.1e} | {rel(ucb, ucbo):.1e} | {rel(rg['x'], ro['x']):.1e} | {rel(rg['value'], ro['value']):.1e} |")
        gp.close()
