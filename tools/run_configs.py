"""All five BASELINE.json configs on one MI355X (C4 itself is bench.py): wall-clock timings -> gpurun_out/configs.json
(copied to profiles/r0N_configs.json)."""
import json, os, re, subprocess, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "64")
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem, synth_candidates
from oracle import oracle_py as oracle
m = sls(); ctx = m.Context(0)
BIN = os.path.join(R, "sequential-line-search_amd", "bin")
out = {}

def wall(f, reps=3):
    f(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

# C1: bayesian_optimization_1d, 20 iterations (GP MAP fit + EI maximisation per iteration)
t0 = time.perf_counter()
p = subprocess.run([os.path.join(BIN, "bayesian_optimization_1d"), "1", "20", "1"], capture_output=True, text=True)
mm = re.search(r"maximizer ([-\d.e]+) maximum ([-\d.e]+)", p.stdout)
out["C1_bayesian_optimization_1d_20_iterations"] = {"wall_s": time.perf_counter() - t0, "maximizer": float(mm.group(1)), "maximum": float(mm.group(2)),
                                                    "true_optimum": [0.852733, 2.273928]}
# C2: N=2048, D=16, ARD-SE: Gram + Cholesky (+ inverse, alpha) + 4096-point predict
D, N, M = 16, 2048, 4096
X, y, theta, b = synth_problem(oracle, D, N); Xs = synth_candidates(oracle, D, M)
gp = m.GP(ctx, X, y, theta, b, 0)
out["C2_gp_fit_predict_N2048_D16_M4096"] = {"fit_ms_wall_incl_upload": wall(lambda: m.GP(ctx, X, y, theta, b, 0).close()),
                                            "predict_ms_wall_incl_pcie": wall(lambda: gp.predict(Xs))}
ctx.prof_enable(True); ctx.prof_reset()
for _ in range(3): m.GP(ctx, X, y, theta, b, 0).close()
out["C2_gp_fit_predict_N2048_D16_M4096"]["fit_device_ms_by_stage"] = {k: ctx.prof_get(k)[0] / 3 for k in ("gram", "potrf", "trtri", "lauum")}
ctx.prof_enable(False)
t0 = time.perf_counter(); ref = oracle.Regressor(X, y, theta, b, kernel=0); t1 = time.perf_counter(); ref.predict_batch(Xs); t2 = time.perf_counter()
out["C2_gp_fit_predict_N2048_D16_M4096"]["cpu_oracle_fit_s"] = t1 - t0
out["C2_gp_fit_predict_N2048_D16_M4096"]["cpu_oracle_predict_s"] = t2 - t1
gp.close()
# C3: sequential_line_search_nd D=32, 30 iterations (PreferenceRegressor MAP + EI acquisition per step)
p = subprocess.run([os.path.join(BIN, "sequential_line_search_nd"), "32", "30", "1"], capture_output=True, text=True)
ms = [float(v) for v in re.findall(r" ms ([-\d.e]+)", p.stdout)]
res = [float(v) for v in re.findall(r"residual ([-\d.e]+)", p.stdout)]
out["C3_sequential_line_search_nd_D32_30_iterations"] = {"ms_per_submit_mean": float(np.mean(ms)), "ms_per_submit_max": float(np.max(ms)),
                                                         # the first submit carries the one-off initialisation (code objects, buffers)
                                                         "ms_per_submit_mean_without_first": float(np.mean(ms[1:])), "ms_per_submit_median": float(np.median(ms)),
                                                         "ms_per_submit_last": ms[-1],
                                                         "residual_first": res[0], "residual_last": res[-1]}
# C5: Matern-5/2 MAP objective + gradient, N=4096, D=128
D, N = 128, 4096
X, y, theta, b = synth_problem(oracle, D, N)
h = m.Nll(ctx, X, 1)
x = np.concatenate([[0.5, 0.005], np.full(D, theta[1])])
k = [0]
def ev():
    k[0] += 1
    xx = x.copy(); xx[2] *= (1 + 1e-3 * k[0])
    h.gp_objective(y, xx)
ms_c5 = wall(ev)
# N^3/3 (potrf) + 2N^3/3 (K^-1 from L) + 2 D N^2 (X G) + N^2 D (Gram) flops per evaluation, against the fp64 MFMA peak
flops_c5 = N ** 3 + 3.0 * D * N * N
out["C5_map_objective_gradient_N4096_D128"] = {"ms_per_evaluation": ms_c5,
                                               "map_eval_roofline": {"bound": "mfma", "flops": flops_c5, "achieved_TFLOPs": flops_c5 / (ms_c5 * 1e-3) / 1e12,
                                                                     "peak_TFLOPs": 78.6, "frac": flops_c5 / (ms_c5 * 1e-3) / 1e12 / 78.6}}
json.dump(out, open(os.path.join(R, "gpurun_out", "configs.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
