"""Randomised sweeps of the path's other entry points (the posterior / acquisition sweep lives in test_gpu_parity.py):
Cholesky / solve / inverse at random sizes against numpy, the GP-MAP objective + gradient against the oracle, the lock-step
maximiser against the oracle's identical algorithm, append-point against refit.  A handful of seeds run in the suite;
SLS_TEST_EXTRA_SEEDS=n adds n more for one-off sweeps (profiles/r03_stress_sweep.log)."""
import os

import numpy as np
import pytest

from util import assert_starts_agree, oracle_end_value_sensitivity, record, sls

pytestmark = pytest.mark.gpu
EXTRA = int(os.environ.get("SLS_TEST_EXTRA_SEEDS", "0"))


@pytest.fixture(scope="module")
def ctx():
    c = sls().Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(6 + EXTRA))
def test_random_cholesky_sizes(ctx, seed):
    """potrf / potrs / potri at a random order (1 .. 1500, every schedule boundary: 128-multiples, 384 = first single-launch
    size) and a random spectrum, against numpy: backward error |L L^T - A| <= 50 N eps |A|, solve residual, A A^-1 = I."""
    rng = np.random.default_rng(5000 + seed)
    N = int(rng.choice([int(rng.integers(1, 1500)), 128 * int(rng.integers(1, 12)), 128 * int(rng.integers(1, 12)) + int(rng.integers(-1, 2))]))
    N = max(N, 1)
    Q = rng.normal(size=(N, N))
    lam = 10.0 ** rng.uniform(-4, 2, N)
    A = (Q * lam) @ Q.T / N + 1e-3 * np.eye(N)
    A = 0.5 * (A + A.T)
    L = ctx.potrf(A)
    eps = np.finfo(float).eps
    nA = np.abs(A).max()
    assert np.all(np.triu(L, 1) == 0)
    assert np.max(np.abs(L @ L.T - A)) <= 50 * N * eps * nA, (N, np.max(np.abs(L @ L.T - A)))
    Lr = np.linalg.cholesky(A)
    assert np.max(np.abs(L - Lr)) <= 1e-9 * np.linalg.cond(A) * eps / 2.2e-16 * np.abs(Lr).max() + 1e-13
    B = rng.normal(size=(N, 2))
    Xs = ctx.potrs(L, B)
    assert np.max(np.abs(A @ Xs - B)) <= 1e3 * N * eps * (nA * np.abs(Xs).max() + np.abs(B).max())
    Ai = ctx.potri(L)
    assert np.array_equal(Ai, Ai.T)
    assert np.max(np.abs(A @ Ai - np.eye(N))) <= 1e3 * N * eps * np.linalg.cond(A)
    record("stress_cholesky", seed=int(seed), N=N, cond=float(np.linalg.cond(A)), backward=float(np.max(np.abs(L @ L.T - A)) / nA))


@pytest.mark.parametrize("seed", range(6 + EXTRA))
def test_random_map_objective_and_gradient(ctx, oracle, seed, monkeypatch):
    """GP-MAP objective (src/gaussian-process-regressor.cpp:36-193) at random shapes, kernels and hyper-parameters -- across
    the N = 128 boundary between the fused one-workgroup kernel and the tiled pipeline -- against the oracle's hoisted form:
    value 1e-9 relative, gradient 1e-6 of its largest component (+ the conditioning of K_y, as for sigma)."""
    rng = np.random.default_rng(6000 + seed)
    D = int(rng.integers(1, 24)); N = int(rng.integers(2, 330)); kernel = int(rng.integers(0, 2))
    X = rng.uniform(0, 1, (D, N))
    y = np.sin(X.sum(axis=0) * rng.uniform(1, 4)) + 0.05 * rng.normal(size=N)
    a = rng.uniform(0.1, 2.0); b = float(10 ** rng.uniform(-5, -1))
    r = rng.uniform(0.2, 1.5, D) * np.sqrt(max(D, 4) / 4.0)
    x = np.concatenate([[a, b], r])
    vo, go = oracle.gp_map_objective(kernel, X, y, x, as_written=False)
    if rng.integers(0, 2):
        monkeypatch.setenv("SLS_NLL_SMALL", "0")
    h = sls().Nll(ctx, X, kernel)
    v, g = h.gp_objective(y, x)
    kappa = (a * N + b) / b
    eps = np.finfo(float).eps
    assert abs(v - vo) <= 1e-9 * abs(vo) + 1e3 * kappa * eps * (1.0 + float(y @ y) / a), (v, vo, kappa)
    gmax = np.abs(go).max()
    assert np.max(np.abs(g - go)) <= 1e-6 * gmax + 1e3 * kappa * eps * gmax, (np.max(np.abs(g - go)) / gmax, kappa)
    record("stress_map", seed=int(seed), D=D, N=N, kernel=kernel, b=b, kappa=float(kappa), value_rel=float(abs(v - vo) / abs(vo)),
           grad_rel=float(np.max(np.abs(g - go)) / gmax))
    h.close()


@pytest.mark.parametrize("seed", range(4 + EXTRA))
def test_random_maximiser_against_the_oracle(ctx, oracle, seed, monkeypatch):
    """sls_acq_maximize (lock-step L-BFGS over the active set, or one wavefront per start) at random shapes against the oracle's
    identical algorithm: every start ends at the oracle's value or has an Armijo test at rounding level of its threshold."""
    rng = np.random.default_rng(7000 + seed)
    D = int(rng.integers(1, 20)); N = int(rng.integers(3, 260)); S = int(rng.integers(1, 160)); kernel = int(rng.integers(0, 2))
    acq = int(rng.integers(0, 2)); n_local = int(rng.integers(2, 30))
    X = rng.uniform(0, 1, (D, N))
    y = np.sin(X.sum(axis=0) * rng.uniform(1, 4)) + 0.05 * rng.normal(size=N)
    theta = np.concatenate([[rng.uniform(0.2, 1.5)], rng.uniform(0.3, 1.2, D) * np.sqrt(max(D, 4) / 4.0)])
    b = float(10 ** rng.uniform(-4, -1))
    starts = rng.uniform(0, 1, (D, S))
    starts[:, ::3] = np.round(starts[:, ::3])
    monkeypatch.setenv("SLS_WAVE_PATH", str(int(rng.integers(0, 2))))
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
    ro = ref.acq_maximize(starts, n_local, acq, 1.3, diag=True)
    rg = gp.acq_maximize(starts, n_local, acq, 1.3)
    # every third start is rounded to a corner: in few dimensions many starts coincide and share one trajectory (a near-threshold
    # Armijo test then shows up several times: seed 4, twelve starts = two trajectories), hence the wider count bounds here
    # a start that does not agree must have a near-threshold Armijo test, or be one whose end value the ORACLE itself does not
    # reproduce under last-place changes of the start / the model (util.oracle_end_value_sensitivity: seeds 76 and 181 of the
    # 200-seed sweep end 1e-5 .. 1e-2 elsewhere under one ulp of the signal variance)
    def ulp_probe(i):
        return oracle_end_value_sensitivity(oracle, X, y, theta, b, kernel, starts, i, n_local, acq, 1.3, ro)
    assert_starts_agree(rg, ro, label=f"stress maximiser seed={seed} D={D} N={N} S={S}", min_frac=0.85, max_divergent=max(2, S // 8),
                        ulp_probe=ulp_probe, atol_scale=1e-10)      # seed 76: an end point at 1e-5 of the largest value, 7.8e-11 off
    assert ro["y_stars"][rg["index"]] >= ro["value"] - 1e-6 * abs(ro["value"]) - 1e-300       # north_star: the chosen maximiser to 1e-6
    gap = abs(rg["value"] - ro["value"])
    if gap > 1e-6 * abs(ro["value"]) + 1e-12:
        # The device's VALUE at its end point against the oracle's: where the posterior itself is ill-conditioned the oracle does not
        # reproduce its own number to 1e-6 under last-place changes (seed 360 of the 400-seed sweep: D = 2, N = 170, b = 6e-4, one
        # step per start, EI = 7.0e-5 in the cancelling tail: oracle band 8.6e-7, device 1.03e-6 away).  Same rule as per start:
        # within twice the oracle's own band.  The choice of the start is held to 1e-6 by the line above, unconditionally.
        sens = ulp_probe(int(rg["index"]))
        record("stress_maximiser_value_band", seed=int(seed), rel_gap=float(gap / abs(ro["value"])), oracle_band=float(sens))
        assert gap <= 2.0 * sens * max(np.abs(ro["y_stars"]).max(), 1e-300), (gap, sens)
    assert np.all((rg["x_stars"] >= 0) & (rg["x_stars"] <= 1))
    gp.close()


@pytest.mark.parametrize("seed", range(4 + EXTRA))
def test_random_append_equals_refit(ctx, oracle, seed):
    """sls_gp_append_point (O(N^2) growth of L, L^-1, K^-1, alpha) against a refit on the grown data, at random sizes around the
    128-padding boundaries."""
    rng = np.random.default_rng(8000 + seed)
    D = int(rng.integers(1, 12)); kernel = int(rng.integers(0, 2))
    N0 = int(rng.choice([int(rng.integers(2, 300)), 128 * int(rng.integers(1, 3)) - int(rng.integers(0, 3))]))
    grow = int(rng.integers(1, 5))
    X = rng.uniform(0, 1, (D, N0 + grow))
    y = np.sin(X.sum(axis=0) * 2.0) + 0.05 * rng.normal(size=N0 + grow)
    theta = np.concatenate([[rng.uniform(0.3, 1.5)], rng.uniform(0.3, 1.2, D) * np.sqrt(max(D, 4) / 4.0)])
    b = float(10 ** rng.uniform(-3, -1))
    gp = sls().GP(ctx, X[:, :N0], y[:N0], theta, b, kernel)
    for k in range(grow):
        gp.append_point(X[:, N0 + k], y[N0 + k])
    full = sls().GP(ctx, X, y, theta, b, kernel)
    Xs = rng.uniform(0, 1, (D, 40))
    m1, s1 = gp.predict(Xs); m2, s2 = full.predict(Xs)
    # the two routes differ in summation order only: agreement to 1e-8 / 1e-7 plus the first-order conditioning term of
    # sigma^2 = a - k^T K_y^-1 k (test_randomised_configurations; seed 154: D = 1, 267 points, sigma down to 4e-3)
    N = N0 + grow
    d_sigma = (theta[0] * N + b) / b * np.finfo(float).eps * theta[0] / (2.0 * np.maximum(s2, 1e-150))
    assert np.max(np.abs(m1 - m2)) <= 1e-8 * max(np.abs(m2).max(), 1e-30)
    assert np.all(np.abs(s1 - s2) <= 1e-7 * max(np.abs(s2).max(), 1e-30) + d_sigma)
    v1 = gp.acq_eval(Xs, 0, 1.0, want_grad=False); v2 = full.acq_eval(Xs, 0, 1.0, want_grad=False)
    assert np.all(np.abs(v1 - v2) <= 1e-6 * max(np.abs(v2).max(), 1e-30) + 0.4 * d_sigma)
    gp.close(); full.close()


@pytest.mark.parametrize("seed", range(4 + EXTRA))
def test_random_preference_objective(ctx, oracle, seed, monkeypatch):
    """PreferenceRegressor's MAP objective (src/preference-regressor.cpp:53-291) at random data: M points in random preference
    tuples of 2-5 indices, goodness values of random scale, with / without the hyper-parameters in the argument, both kernels,
    fused small kernel or tiled pipeline -- value 1e-9, gradient 1e-6 of its largest component (+ conditioning term)."""
    rng = np.random.default_rng(9000 + seed)
    D = int(rng.integers(1, 16)); M = int(rng.integers(3, 160)); kernel = int(rng.integers(0, 2)); use_map = bool(rng.integers(0, 2))
    X = rng.uniform(0, 1, (D, M))
    prefs = []
    for _ in range(int(rng.integers(1, max(2, M // 2)))):
        n = int(rng.integers(2, min(6, M + 1)))
        prefs.append([int(i) for i in rng.choice(M, size=n, replace=False)])
    yv = rng.normal(0, 10 ** rng.uniform(-2.5, -1.0), M)
    a = rng.uniform(0.2, 1.0); b = float(10 ** rng.uniform(-4, -1.5))
    x = np.concatenate([yv, [a, b], rng.uniform(0.3, 0.9, D)]) if use_map else yv
    kw = {} if use_map else dict(a=a, b=b, r=float(rng.uniform(0.3, 0.9)))
    vo, go = oracle.pref_objective(kernel, X, prefs, x, use_map=use_map, **kw)
    if rng.integers(0, 2):
        monkeypatch.setenv("SLS_NLL_SMALL", "0")
    h = sls().Nll(ctx, X, kernel)
    v, g = h.pref_objective(prefs, x, use_map=use_map, **kw)
    kappa = (a * M + b) / b
    eps = np.finfo(float).eps
    assert abs(v - vo) <= 1e-9 * abs(vo) + 1e3 * kappa * eps * (1.0 + float(yv @ yv) / a), (v, vo)
    gmax = np.abs(go).max()
    assert np.max(np.abs(g - go)) <= 1e-6 * gmax + 1e3 * kappa * eps * gmax, (np.max(np.abs(g - go)) / gmax, kappa)
    record("stress_pref", seed=int(seed), D=D, M=M, kernel=kernel, use_map=use_map, value_rel=float(abs(v - vo) / abs(vo)),
           grad_rel=float(np.max(np.abs(g - go)) / gmax))
    h.close()


@pytest.mark.parametrize("seed", range(3 + EXTRA // 4))
def test_random_sharded_maximisation_is_bit_identical(oracle, seed):
    """sls_multi_acq_maximize over 2-5 logical shards of the one GPU against the single-device call at random shapes: winner index,
    value and point bit for bit, and the same evaluation count (every start retires at the same evaluation)."""
    rng = np.random.default_rng(9500 + seed)
    m = sls()
    D = int(rng.integers(1, 12)); N = int(rng.integers(3, 400)); S = int(rng.integers(1, 700)); kernel = int(rng.integers(0, 2))
    n_local = int(rng.integers(2, 16)); shards = int(rng.integers(2, 6))
    X = rng.uniform(0, 1, (D, N))
    y = np.sin(X.sum(axis=0) * 2.5) + 0.05 * rng.normal(size=N)
    theta = np.concatenate([[rng.uniform(0.2, 1.5)], rng.uniform(0.3, 1.2, D) * np.sqrt(max(D, 4) / 4.0)])
    b = float(10 ** rng.uniform(-4, -1))
    starts = rng.uniform(0, 1, (D, S))
    c = m.Context(0)
    gp = m.GP(c, X, y, theta, b, kernel)
    one = gp.acq_maximize(starts, n_local)
    issued = gp.last_stats()["evals_issued"]
    multi = m.Multi([0] * shards)
    mgp = m.MultiGP(multi, X, y, theta, b, kernel)
    r = mgp.acq_maximize(starts, n_local)
    assert r["index"] == one["index"] and r["value"] == one["value"] and np.array_equal(r["x"], one["x"]), (seed, shards, S)
    assert r["evals_issued"] == issued
    mgp.close(); multi.close(); gp.close(); c.close()


@pytest.mark.parametrize("N", [2049, 3001, 4999])
def test_large_ragged_orders(ctx, N):
    """Orders that are not multiples of the 128 tile, in the range of the single-launch dataflow Cholesky with many tiles per
    owner (identity padding inside the last tile row): factor, solve and inverse against numpy's residual bounds."""
    rng = np.random.default_rng(N)
    Q = rng.normal(size=(N, 64))
    A = Q @ Q.T / 64 + np.diag(10.0 ** rng.uniform(-3, 0, N))
    L = ctx.potrf(A)
    eps = np.finfo(float).eps
    nA = np.abs(A).max()
    assert np.all(np.triu(L, 1) == 0)
    assert np.max(np.abs(L @ L.T - A)) <= 50 * N * eps * nA
    B = rng.normal(size=(N, 3))
    Xs = ctx.potrs(L, B)
    assert np.max(np.abs(A @ Xs - B)) <= 1e3 * N * eps * (nA * np.abs(Xs).max() + np.abs(B).max())
    Ai = ctx.potri(L)
    assert np.array_equal(Ai, Ai.T)
    assert np.max(np.abs(A @ Ai - np.eye(N))) <= 1e3 * N * eps * np.linalg.cond(A)


@pytest.mark.parametrize("seed", range(3 + EXTRA // 4))
def test_random_gp_map_fit_ends_at_a_bounded_stationary_point(ctx, seed):
    """GaussianProcessRegressor(X, y, kernel) -- PerformMapEstimation (src/gaussian-process-regressor.cpp:274-299) through the
    restated C++ class -- on random data: the fit must return finite hyper-parameters inside the reference's box, not below the
    DIRECT point or the prior medians, with a projected gradient (in the log-parameters the local phase runs in, from an
    independent device evaluation) of at most 1e-4 |objective| unless the 1000-evaluation cap of the local phase was reached."""
    import ctypes as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sls().lib()
    host = C.CDLL(os.path.join(root, "sequential-line-search_amd", "libsequential-line-search.so"))
    dp = C.POINTER(C.c_double)
    host.slsh_gp_map_fit.argtypes = [dp, C.c_int, C.c_int, dp, C.c_int, dp, dp]
    host.slsh_last_error.restype = C.c_char_p
    rng = np.random.default_rng(9700 + seed)
    D = int(rng.integers(1, 9)); N = int(rng.integers(3, 160)); kernel = int(rng.integers(0, 2))
    X = np.asfortranarray(rng.uniform(0, 1, (D, N)))
    y = np.ascontiguousarray(np.sin(X.sum(axis=0) * rng.uniform(1, 5)) * rng.uniform(0.2, 2.0) + 10 ** rng.uniform(-3, -1) * rng.normal(size=N))
    x = np.zeros(D + 2); st = np.zeros(8)
    rc = host.slsh_gp_map_fit(X.ctypes.data_as(dp), D, N, y.ctypes.data_as(dp), kernel, x.ctypes.data_as(dp), st.ctypes.data_as(dp))
    assert rc == 0, host.slsh_last_error()
    final, direct, prior, evals_local = st[0], st[1], st[2], int(st[4])
    lo, hi = np.log(1e-8), np.log(50.0)
    assert np.all(np.isfinite(x)) and np.all(x > 0) and np.all(np.log(x) >= lo - 1e-9) and np.all(np.log(x) <= hi + 1e-9)
    assert final >= direct - 1e-9 * max(1.0, abs(final)) and final >= prior - 1e-9 * max(1.0, abs(final))
    h = sls().Nll(ctx, X, kernel)
    v, g = h.gp_objective(y, x)
    h.close()
    assert abs(v - final) <= 1e-9 * max(1.0, abs(v))
    gz = g * x
    z = np.log(x)
    gz[(z <= lo + 1e-12) & (gz < 0)] = 0.0
    gz[(z >= hi - 1e-12) & (gz > 0)] = 0.0
    record("stress_map_fit", seed=int(seed), D=D, N=N, kernel=kernel, objective=float(final), pg_over_obj=float(np.max(np.abs(gz)) / max(1.0, abs(v))),
           evals_local=evals_local)
    assert evals_local >= 1000 or np.max(np.abs(gz)) <= 1e-4 * max(1.0, abs(v)), (np.max(np.abs(gz)), v, evals_local)
