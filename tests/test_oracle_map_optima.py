"""CPU: the oracle's two MAP objectives (oracle/sls_oracle.c restating src/gaussian-process-regressor.cpp:36-193 and
src/preference-regressor.cpp:53-291) against the independent numpy / LAPACK implementation behind tests/golden/map_optima.npz
(make_map_optima.py) AT the optima scipy found: same value (1e-9) and a vanishing projected gradient there, in both the
hoisted and the as-written formulation of the oracle.  N up to 300, D up to 32: far beyond the mpmath fixtures (N <= 10)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Z = np.load(os.path.join(ROOT, "tests", "golden", "map_optima.npz"))
Z3 = np.load(os.path.join(ROOT, "tests", "golden", "map_optima_c3.npz"))     # config 3's shape: D = 32, M = 91 / 40 (round 5)


def case(name):
    z = Z3 if name + "/X" in Z3.files else Z
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


@pytest.mark.parametrize("name", [str(n) for n in Z["gp_cases"]])
def test_gp_map_objective_at_scipy_optima(oracle, name):
    c = case(name)
    X, y, kind = np.asfortranarray(c["X"]), c["y"], int(c["kernel"])
    N = X.shape[1]
    for k, x in enumerate(c["local_x"]):
        v, g = oracle.gp_map_objective(kind, X, y, x)
        scale = max(1.0, abs(float(c["local_values"][k])))
        assert abs(v - float(c["local_values"][k])) <= 1e-9 * scale, (k, v, c["local_values"][k])
        gz = g * x                                       # gradient in the log-parameters the optimisers ran in
        z = np.log(x)
        gz[(z <= np.log(1e-8) + 1e-12) & (gz < 0)] = 0.0
        gz[(z >= np.log(50.0) - 1e-12) & (gz > 0)] = 0.0
        assert np.max(np.abs(gz)) <= 2e-4 * scale, (k, np.max(np.abs(gz)))
    if N <= 90:                                          # the reference's own formulation: tensor of dK/dtheta + traces (O(D N^3))
        va, ga = oracle.gp_map_objective(kind, X, y, c["x_opt"], as_written=True)
        vh, gh = oracle.gp_map_objective(kind, X, y, c["x_opt"])
        np.testing.assert_allclose(va, vh, rtol=1e-10)
        np.testing.assert_allclose(ga, gh, rtol=1e-6, atol=1e-6 * max(1.0, np.abs(gh).max()))


@pytest.mark.parametrize("name", [str(n) for n in Z["pref_cases"]] + [str(n) for n in Z3["pref_cases"]])
def test_pref_objective_at_scipy_optima(oracle, name):
    c = case(name)
    X, kind, use_map = np.asfortranarray(c["X"]), int(c["kernel"]), bool(int(c["use_map"]))
    D, M = X.shape
    offs = c["offsets"]
    prefs = [[int(i) for i in c["prefs_flat"][offs[p]:offs[p + 1]]] for p in range(len(offs) - 1)]
    x = c["x_opt"]
    v, g = oracle.pref_objective(kind, X, prefs, x, use_map=use_map)
    scale = max(1.0, abs(float(c["value"])))
    assert abs(v - float(c["value"])) <= 1e-9 * scale, (v, float(c["value"]))
    gz = np.array(g, dtype=float)
    if use_map:
        gz[M:] *= x[M:]
    inside = np.ones(len(x), bool)
    inside[:M] = np.abs(x[:M]) < 10.0 - 1e-9
    assert np.max(np.abs(gz[inside])) <= 2e-4 * scale, np.max(np.abs(gz[inside]))
