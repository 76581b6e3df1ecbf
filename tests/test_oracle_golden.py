"""The oracle (oracle/sls_oracle.c) against the independent golden fixtures
(tests/golden/fixtures.json: mpmath 50-digit + sklearn).  CPU only."""
import numpy as np
import pytest

RTOL = 1e-10


def close(a, b, rtol=RTOL, atol=1e-13):
    np.testing.assert_allclose(np.asarray(a, dtype=float), np.asarray(b, dtype=float), rtol=rtol, atol=atol)


def test_kernel_scalars(fixtures, oracle):
    for c in fixtures["kernel_scalars"]:
        close(oracle.kernel(c["kernel"], c["xa"], c["xb"], c["theta"]), c["k"])
        close(oracle.kernel_theta_derivative(c["kernel"], c["xa"], c["xb"], c["theta"]), c["dtheta"])
        close(oracle.kernel_first_arg_derivative(c["kernel"], c["xa"], c["xb"], c["theta"]), c["dx"])


def test_normal_and_lognormal(fixtures, oracle):
    lib = oracle.lib()
    for c in fixtures["ei_scalars"]:
        u = (c["mu"] - c["mu_best"]) / c["sigma"]
        close(lib.slso_norm_cdf(u), c["cdf"], rtol=1e-9, atol=1e-300)
        close(lib.slso_norm_pdf(u), c["pdf"], rtol=1e-9, atol=1e-300)
    for c in fixtures["lognormal"]:
        close(lib.slso_log_lognormal(c["x"], c["mu"], c["sigma2"]), c["logpdf"])
        close(lib.slso_log_lognormal_derivative(c["x"], c["mu"], c["sigma2"]), c["dlogpdf"])


def test_btl(fixtures, oracle):
    for c in fixtures["btl"]:
        close(oracle.btl(c["f"], c["scale"]), c["btl"])
        close(oracle.btl_derivative(c["f"], c["scale"]), c["dbtl"], rtol=1e-9)


@pytest.mark.parametrize("reg_type", [0, 1])
def test_gp_pipeline_mpmath(fixtures, oracle, reg_type):
    """posterior, gradients, EI/UCB of both regressor classes (as-written and hoisted) vs 50-digit mpmath."""
    for c in fixtures["gp_pipelines"]:
        X = np.array(c["X"]); Xs = np.array(c["Xs"])
        r = oracle.Regressor(X, c["y"], c["theta"], c["b"], kernel=c["kernel"], reg_type=reg_type)
        tol = dict(rtol=2e-7, atol=1e-9)   # kappa(K) ~ 1e5..1e6 amplifies fp64 rounding
        close(oracle.calc_large_ky(c["kernel"], X, c["theta"], c["b"]), c["K"])
        idx, xb = r.predict_maximum_point_from_data()
        assert idx == c["best_index"]
        mu_b, sg_b = r.predict_batch(Xs)
        dmu_b, dsg_b = r.predict_grad_batch(Xs)
        ei_b, dei_b = r.acq_eval_batch(Xs, oracle.ACQ_EI)
        ucb_b, ducb_b = r.acq_eval_batch(Xs, oracle.ACQ_UCB, ucb_h=2.0)
        close(mu_b, c["mu"], **tol); close(sg_b, c["sigma"], **tol)
        close(dmu_b.T, c["dmu"], **tol); close(dsg_b.T, c["dsigma"], rtol=2e-6, atol=1e-8)
        close(ei_b, c["ei"], rtol=2e-6, atol=1e-10); close(dei_b.T, c["dei"], rtol=2e-6, atol=1e-9)
        close(ucb_b, c["ucb"], **tol); close(ducb_b.T, c["ducb"], rtol=2e-6, atol=1e-8)
        for m in range(Xs.shape[1]):
            x = Xs[:, m]
            close(r.predict_mu(x), c["mu"][m], **tol)
            close(r.predict_sigma(x), c["sigma"][m], **tol)
            close(r.predict_mu_derivative(x), c["dmu"][m], **tol)
            close(r.predict_sigma_derivative(x), c["dsigma"][m], rtol=2e-6, atol=1e-8)
            close(r.acq_value_as_written(x, oracle.ACQ_EI), c["ei"][m], rtol=2e-6, atol=1e-10)
            close(r.acq_derivative_as_written(x, oracle.ACQ_EI), c["dei"][m], rtol=2e-6, atol=1e-9)
            close(r.acq_value_as_written(x, oracle.ACQ_UCB, 2.0), c["ucb"][m], **tol)
            close(r.acq_derivative_as_written(x, oracle.ACQ_UCB, 2.0), c["ducb"][m], rtol=2e-6, atol=1e-8)


def test_linear_algebra(fixtures, oracle):
    for c in fixtures["gp_pipelines"]:
        K = np.array(c["K"])
        L, info = oracle.cholesky(K)
        assert info == 0
        close(L @ L.T, K, rtol=1e-13)
        assert np.all(np.triu(L, 1) == 0)
        close(oracle.logdet_from_chol(L), c["logdet"], rtol=1e-11)
        close(oracle.spd_inverse_from_chol(L), c["Kinv"], rtol=1e-8, atol=1e-8)
        inv, info = oracle.lu_inverse(K)
        assert info == 0
        close(inv, c["Kinv"], rtol=1e-8, atol=1e-8)
        close(oracle.chol_solve(L, np.array(c["y"])), c["alpha"], rtol=1e-8, atol=1e-9)
    rng = np.random.default_rng(3)
    A = rng.normal(size=(300, 300)); A = A @ A.T + 300 * np.eye(300)   # exercises the blocked path (NB = 64)
    L, info = oracle.cholesky(A)
    assert info == 0
    close(L, np.linalg.cholesky(A), rtol=1e-11)


def test_sklearn_third_opinion(fixtures, oracle):
    for c in fixtures["sklearn_gp"]:
        r = oracle.Regressor(np.array(c["X"]), c["y"], c["theta"], c["b"], kernel=c["kernel"])
        mu, sg = r.predict_batch(np.array(c["Xs"]))
        close(mu, c["mu"], rtol=1e-7, atol=1e-9)
        close(sg, c["sigma"], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("as_written", [True, False])
def test_gp_map_objective(fixtures, oracle, as_written):
    for c in fixtures["gp_map"]:
        v, g = oracle.gp_map_objective(c["kernel"], np.array(c["X"]), c["y"], c["x"], as_written=as_written)
        close(v, c["value"], rtol=1e-9)
        close(g, c["grad"], rtol=1e-6, atol=1e-7)


def test_pref_objective(fixtures, oracle):
    for c in fixtures["pref_objective"]:
        v, g = oracle.pref_objective(c["kernel"], np.array(c["X"]), c["prefs"], c["x"], use_map=c["use_map"])
        close(v, c["value"], rtol=1e-9)
        close(g, c["grad"], rtol=1e-6, atol=1e-6)


def test_as_written_equals_hoisted(oracle):
    """The O(N^3)-per-evaluation reference call structure and the hoisted batched form agree."""
    rng = np.random.default_rng(11)
    for kernel in (0, 1):
        for reg_type in (0, 1):
            D, N, M = 5, 48, 12
            X = rng.uniform(0, 1, (D, N)); y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * rng.normal(size=N)
            theta = np.concatenate([[0.5], np.full(D, 0.5)])
            r = oracle.Regressor(X, y, theta, 0.005, kernel=kernel, reg_type=reg_type)
            Xs = rng.uniform(0, 1, (D, M))
            ei, dei = r.acq_eval_batch(Xs)
            for m in range(M):
                close(r.acq_value_as_written(Xs[:, m]), ei[m], rtol=1e-7, atol=1e-12)
                close(r.acq_derivative_as_written(Xs[:, m]), dei[:, m], rtol=1e-6, atol=1e-10)


def test_gradients_by_finite_differences(oracle):
    rng = np.random.default_rng(5)
    D, N = 4, 30
    X = rng.uniform(0, 1, (D, N)); y = np.sin(3 * X.sum(axis=0))
    theta = np.array([0.5, 0.4, 0.5, 0.6, 0.7])
    for kernel in (0, 1):
        r = oracle.Regressor(X, y, theta, 0.01, kernel=kernel)
        x = rng.uniform(0.1, 0.9, D); h = 1e-6
        for f, df in ((r.predict_mu, r.predict_mu_derivative), (r.predict_sigma, r.predict_sigma_derivative),
                      (r.acq_value_as_written, r.acq_derivative_as_written)):
            g = df(x)
            for d in range(D):
                e = np.zeros(D); e[d] = h
                close((f(x + e) - f(x - e)) / (2 * h), g[d], rtol=2e-5, atol=1e-8)


def test_multistart_maximizer_properties(oracle):
    """1-D BO scenario of demos/bayesian_optimization_1d/core.cpp:70-73."""
    rng = np.random.default_rng(1)
    X = rng.uniform(0, 1, (1, 8)); y = 1.0 - 1.5 * X[0] * np.sin(13.0 * X[0])
    r = oracle.Regressor(X, y, [0.5, 0.15], 1e-4, kernel=1)
    starts = rng.uniform(0, 1, (1, 64))
    res = r.acq_maximize(starts, 30)
    v0 = r.acq_eval_batch(starts, want_grad=False)
    assert np.all(res["y_stars"] >= v0 - 1e-15)             # monotone: never worse than the start
    assert res["index"] == int(np.argmax(res["y_stars"]))   # first maximum
    close(res["value"], res["y_stars"].max())
    assert np.all((res["x_stars"] >= 0) & (res["x_stars"] <= 1))
    grid = np.linspace(0, 1, 4001)[None, :]
    vg = r.acq_eval_batch(grid, want_grad=False)
    assert res["value"] >= vg.max() * (1 - 1e-3)            # found the global EI maximiser on [0,1]
    close(r.acq_eval_batch(res["x"][:, None], want_grad=False)[0], res["value"], rtol=1e-12)
