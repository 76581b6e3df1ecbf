"""Where NLopt's relative stopping tests pick the ANSWER: the reference's default maximiser branch is DIRECT followed by ONE L-BFGS
(src/acquisition-function.cpp:155-165), and every search of the reference runs under nloptutil::solve's ftol_rel = xtol_rel = 1e-6
(SURVEY.md Appendix A).  Under a multi-start maximum a start that stops on one short accepted step is harmless; here the point the
single local search returns IS the next query.  C3's shapes: D = 32, N = 10 .. 91 (demos/sequential_line_search_nd/main.cpp:86-91),
50 D DIRECT evaluations, 10 D local ones (src/sequential-line-search.cpp:71-72 as restated in host/sequential-line-search.cpp).
"""
import ctypes as C
import importlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sequential-line-search_amd")
pytestmark = pytest.mark.gpu


def sls():
    return importlib.import_module("sequential-line-search_amd")


@pytest.fixture(scope="module")
def host():
    sls().lib()
    lib = C.CDLL(os.path.join(PKG, "libsequential-line-search.so"))
    dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint)
    lib.slsh_find_next_point_direct.argtypes = [dp, C.c_int, C.c_int, up, ip, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                                C.c_double, C.c_int, C.c_uint, C.c_uint, C.c_double, C.c_double, dp, dp]
    lib.slsh_last_error.restype = C.c_char_p
    return lib


def line_search_data(rng, D, M):
    """Data as a line search leaves it: triples (chosen point, the two ends of its slider), the chosen one preferred."""
    X = np.empty((D, M))
    prefs = []
    best = rng.uniform(0.2, 0.8, D)
    i = 0
    while i < M:
        if M - i < 3:                                            # what remains: single points against the current best's index
            X[:, i] = np.clip(best + 0.3 * rng.normal(size=D), 0, 1)
            prefs.append([0, i] if i > 0 else [0])
            i += 1
            continue
        e0 = np.clip(best + 0.4 * rng.normal(size=D), 0, 1)
        e1 = np.clip(best - 0.4 * rng.normal(size=D), 0, 1)
        t = rng.uniform(0.1, 0.9)
        X[:, i], X[:, i + 1], X[:, i + 2] = (1 - t) * e0 + t * e1, e0, e1
        prefs.append([i, i + 1, i + 2])
        best = X[:, i]
        i += 3
    return np.asfortranarray(X), [p for p in prefs if len(p) >= 2]


def find(host, X, prefs, tol, num_global, num_local):
    D, M = X.shape
    flat = np.array([v for p in prefs for v in p], dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum([len(p) for p in prefs])]).astype(np.int32)
    x, v = np.zeros(D), np.zeros(1)
    dp = C.POINTER(C.c_double)
    rc = host.slsh_find_next_point_direct(X.ctypes.data_as(dp), D, M, flat.ctypes.data_as(C.POINTER(C.c_uint)),
                                          offs.ctypes.data_as(C.POINTER(C.c_int)), len(prefs), 0.5, 0.5, 0.005, 0.25, 0.01, 1,
                                          num_global, num_local, tol, tol, x.ctypes.data_as(dp), v.ctypes.data_as(dp))
    assert rc == 0, host.slsh_last_error()
    return x, float(v[0])


def test_single_start_branch_early_stop_against_run_to_cap(host):
    """60 seeds at C3's shapes.  For each: the point DIRECT -> L-BFGS returns with the reference's tolerances (1e-6 / 1e-6) against
    the same search run to its cap of 10 D evaluations (tolerances 0).  DIRECT is deterministic, so both local searches start from
    the same point and the early one is a prefix of the other: the capped run can only end higher.  MEASURED (profiles/
    r06_early_stop_single_start.json): what the early stop leaves on the table, relative to the capped run's value.  NLopt's tests
    look at ONE accepted step: they bound the last step, not the distance to the optimum, so the loss is not bounded by 1e-6 --
    the bound asserted here is the measured one with headroom, and the distribution is recorded."""
    D = 32
    sizes = [10, 19, 28, 37, 46, 55, 64, 73, 82, 91]
    rows = []
    for seed in range(60):
        rng = np.random.default_rng(9000 + seed)
        M = sizes[seed % len(sizes)]
        X, prefs = line_search_data(rng, D, M)
        x_t, v_t = find(host, X, prefs, 1e-6, 50 * D, 10 * D)
        x_c, v_c = find(host, X, prefs, 0.0, 50 * D, 10 * D)
        assert np.all(x_t >= 0) and np.all(x_t <= 1) and np.isfinite(v_t) and np.isfinite(v_c)
        scale = max(abs(v_c), 1e-300)
        rows.append({"seed": seed, "M": M, "value_tol": v_t, "value_cap": v_c, "loss_rel": (v_c - v_t) / scale,
                     "dx_inf": float(np.max(np.abs(x_t - x_c)))})
        assert v_c >= v_t - 1e-9 * scale, rows[-1]                # a prefix cannot end above its continuation
    loss = np.array([r["loss_rel"] for r in rows])
    out = {"D": D, "seeds": len(rows), "loss_rel_max": float(loss.max()), "loss_rel_median": float(np.median(loss)),
           "loss_rel_p90": float(np.quantile(loss, 0.9)), "n_loss_above_1e-5": int((loss > 1e-5).sum()),
           "n_loss_above_1e-3": int((loss > 1e-3).sum()), "dx_inf_max": float(max(r["dx_inf"] for r in rows)), "rows": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "early_stop_single_start.json"), "w") as f:
        json.dump(out, f, indent=1)
    assert loss.max() <= EARLY_STOP_LOSS_BOUND, out["loss_rel_max"]


# measured over the 60 seeds (MI355X, round 6): max 5.3e-6, median 7e-8, 90 % below 3.6e-7 of the capped run's value; the largest
# move of the returned point 0.10 in one coordinate (a flat ridge of the acquisition function at N = 10), typically 1e-3
EARLY_STOP_LOSS_BOUND = 2e-5
