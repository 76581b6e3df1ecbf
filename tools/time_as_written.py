"""CPU baseline in the reference's *as-written* call structure, extrapolated to the C4 size.

One acquisition evaluation as the reference's NLopt callback performs it (`src/acquisition-function.cpp:38-56`: derivative
then value, each recomputing `PredictMaximumPointFromData` = N x PredictMu = O(N^3)) is timed single-threaded through the
oracle's as-written entry points at N in {128, 256, 512, 1024}, D = 64. A power law t = c N^p is fitted to the three
largest sizes and evaluated at N = 8192. The multi-start loop of the reference is embarrassingly parallel over `nproc`
threads, so the whole-machine rate is cores / t. The result is an EXTRAPOLATION and is labelled so.

Writes profiles/r03_cpu_as_written.json (round 3: run on the GPU box, whose core count it records). Test/measurement infrastructure only (uses oracle/).
"""
import json, os, sys, time
import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from oracle import oracle_py as orc

os.environ.setdefault("OMP_NUM_THREADS", "1")
D = 64
sizes = [128, 256, 512, 1024]
if len(sys.argv) > 1:
    sizes = [int(s) for s in sys.argv[1].split(",")]
rows = []
for N in sizes:
    rng = np.random.default_rng(1234 + N)
    X = np.asfortranarray(rng.uniform(0, 1, (D, N)))
    y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * rng.standard_normal(N)
    theta = np.concatenate([[0.5], np.full(D, 0.5 * np.sqrt(D / 8.0))])
    ref = orc.Regressor(X, y, theta, 0.005, kernel=1)
    x = rng.uniform(0, 1, D)
    reps = max(1, int(2e8 / N ** 3))
    t0 = time.perf_counter()
    for _ in range(reps):
        ref.acq_derivative_as_written(x)
        ref.acq_value_as_written(x)
    dt = (time.perf_counter() - t0) / reps
    rows.append({"N": N, "s_per_eval": dt, "reps": reps})
    print("N=%5d  %.4f s per as-written evaluation (value+gradient), %d reps" % (N, dt, reps), flush=True)

fit = rows[-3:]
p, logc = np.polyfit(np.log([r["N"] for r in fit]), np.log([r["s_per_eval"] for r in fit]), 1)
t8192 = float(np.exp(logc) * 8192 ** p)
cores = os.cpu_count()
out = {
    "what": "reference as-written acquisition evaluation (value + gradient through the NLopt-callback structure), "
            "oracle C restatement, 1 thread per evaluation, D=64, Matern-5/2, EI",
    "measured": rows,
    "fit": {"exponent": float(p), "sizes": [r["N"] for r in fit]},
    "extrapolated": True,
    "N8192_s_per_eval_1thread": t8192,
    "N8192_evals_per_s_all_cores": cores / t8192,
    "cores": cores,
    "C4_step_evals": 65536 * 50,
    "C4_step_seconds_all_cores": 65536 * 50 * t8192 / cores,
}
print(json.dumps(out))
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(R, "gpurun_out", "cpu_as_written.json"), "w"), indent=1)
