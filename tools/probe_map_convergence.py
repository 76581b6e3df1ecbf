"""When would NLopt's relative tolerances (nlopt-util's defaults ftol_rel = xtol_rel = 1e-6) stop the MAP fit and
the local phase of config C3?  Runs the C3 scenario (D = 32, 30 line searches) through the pybind11 module, dumps the data of the
last submits (DampData: X.csv, D.csv), and replays the preference MAP fit on the device with evaluation caps 1 .. 100, recording the
objective after every evaluation: the first k at which an accepted step changes f by less than 1e-6 (|f_new| + |f_old|) / 2.
    gpurun -- 'python tools/probe_map_convergence.py'"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sequential-line-search_amd"))
import importlib

sls = importlib.import_module("sequential-line-search_amd")
import pySequentialLineSearch as py


def objective(x):
    return float(np.exp(-np.sum((x - 0.4) ** 2)))


def main():
    D = 32
    py.set_random_seed(1)
    opt = py.SequentialLineSearchOptimizer(num_dims=D, use_map_hyperparams=True)
    opt.set_hyperparams(kernel_signal_var=0.50, kernel_length_scale=0.50, noise_level=0.001, kernel_hyperparams_prior_var=0.10, btl_scale=0.01)
    ts = np.arange(1001) / 1000.0
    ctx = sls.Context()
    for it in range(30):
        vals = [objective(opt.calc_point_from_slider_position(float(t))) for t in ts]
        opt.submit_feedback_data(float(ts[int(np.argmax(vals))]))
        if it not in (9, 19, 29):
            continue
        d = tempfile.mkdtemp()
        opt.damp_data(d + "/")
        X = np.loadtxt(os.path.join(d, "X.csv"), delimiter=",", ndmin=2)
        prefs = [[int(v) for v in line.strip().split(",") if v != ""] for line in open(os.path.join(d, "D.csv")) if line.strip()]
        N = X.shape[1]
        n = N + 2 + D
        lower = np.full(n, -10.0); upper = np.full(n, 10.0)
        lower[N:] = np.log(1e-8); upper[N:] = np.log(10.0)
        z0 = np.zeros(n); z0[N] = np.log(0.5); z0[N + 1] = np.log(0.001); z0[N + 2:] = np.log(0.5)
        h = sls.Nll(ctx, X, 1)
        f = []
        for k in range(1, 101):
            r = h.pref_map_fit(prefs, z0, lower, upper, k, use_map=True, a=0.5, r=0.5, b=0.001, prior_var=0.10, btl_scale=0.01)
            f.append(r["value"])
        h.close()
        # the local phase of the acquisition maximiser on the fitted model: 8 starts, value after k = 1 .. 320 evaluations each
        z = r["z"]
        gp = sls.GP(ctx, X, z[:N], np.concatenate([[np.exp(z[N])], np.exp(z[N + 2:])]), float(np.exp(z[N + 1])), 1)
        gp.set_sigma_mode(sls.SIGMA_CHOLESKY_SOLVE)
        rng = np.random.default_rng(it)
        stops = []
        for s_ in range(8):
            start = rng.uniform(0, 1, (D, 1))
            fv = np.array([gp.acq_maximize(start, k)["value"] for k in range(1, 321)])
            st_, prev_ = None, fv[0]
            for k in range(1, 320):
                if fv[k] != prev_:
                    if abs(fv[k] - prev_) < 1e-6 * 0.5 * (abs(fv[k]) + abs(prev_)) and st_ is None:
                        st_ = k + 1
                    prev_ = fv[k]
            last_change = int(np.max(np.nonzero(np.diff(fv))[0]) + 2) if np.any(np.diff(fv)) else 1
            stops.append((st_, last_change, float(fv[-1])))
        gp.close()
        print("   local phase, 8 random starts: (evaluation at which ftol_rel 1e-6 would stop, last evaluation that still changed the value, final value):", stops)
        f = np.array(f)
        # f[k-1] = best objective after k evaluations (the optimiser keeps the best accepted point): accepted steps are the increases
        stop = None
        prev = f[0]
        for k in range(1, 100):
            if f[k] != prev:
                rel = abs(f[k] - prev) / (0.5 * (abs(f[k]) + abs(prev)))
                if rel < 1e-6 and stop is None:
                    stop = k + 1
                prev = f[k]
        print(f"submit {it + 1}: N {N}  f after 10/25/50/75/100 evaluations: {f[9]:.6f} {f[24]:.6f} {f[49]:.6f} {f[74]:.6f} {f[99]:.6f}   "
              f"ftol_rel 1e-6 would stop at evaluation {stop}   (f100 - f50) / |f100| = {(f[99] - f[49]) / abs(f[99]):.2e}")


if __name__ == "__main__":
    main()
