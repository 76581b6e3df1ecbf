"""The oracle against the scipy/LAPACK pipelines at N = 130, 300, 512, 2048 (tests/golden/scipy_pipelines.npz).

These sizes run the oracle's BLOCKED Cholesky (NB = 64 panels), its blocked SPD inverse and its 16-wide batched
evaluation blocks -- the code the full-size GPU parity tests lean on -- which the N <= 10 mpmath fixtures never reach.
CPU only."""
import numpy as np
import pytest

CASES = [f"k{k}_N{n}" for n in (130, 300, 512, 2048) for k in (0, 1)]


def close(a, b, rtol, atol=0.0):
    np.testing.assert_allclose(np.asarray(a, dtype=float), np.asarray(b, dtype=float), rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("reg_type", [0, 1])
def test_oracle_pipeline_against_scipy(scipy_cases, oracle, name, reg_type):
    c = scipy_cases[name]
    if reg_type == 1 and c["X"].shape[1] > 512:
        pytest.skip("the PreferenceRegressor-style predictive state shares the Cholesky path already covered at N <= 512")
    X, y, theta, b, Xs, kernel = c["X"], c["y"], c["theta"], float(c["b"]), c["Xs"], int(c["kernel"])
    h = scipy_cases["_ucb_h"]
    K = oracle.calc_large_ky(kernel, X, theta, b)
    L, info = oracle.cholesky(K)
    assert info == 0
    rows = c["rows"]
    close(np.diag(L), c["L_diag"], rtol=1e-10)
    close(L[rows], c["L_rows"], rtol=1e-8, atol=1e-12)
    close(oracle.logdet_from_chol(L), c["logdet"], rtol=1e-12)
    Kinv = oracle.spd_inverse_from_chol(L)
    scale = np.abs(c["Kinv_diag"]).max()
    close(np.diag(Kinv), c["Kinv_diag"], rtol=1e-8)
    close(Kinv[rows], c["Kinv_rows"], rtol=1e-7, atol=1e-9 * scale)
    close(oracle.chol_solve(L, y), c["alpha"], rtol=1e-7, atol=1e-9 * np.abs(c["alpha"]).max())
    r = oracle.Regressor(X, y, theta, b, kernel=kernel, reg_type=reg_type)
    assert r.predict_maximum_point_from_data()[0] == int(c["best_index"])
    mu, sg = r.predict_batch(Xs)
    dmu, dsg = r.predict_grad_batch(Xs)
    ei, dei = r.acq_eval_batch(Xs, oracle.ACQ_EI)
    ucb, ducb = r.acq_eval_batch(Xs, oracle.ACQ_UCB, ucb_h=h)
    close(mu, c["mu"], rtol=1e-8, atol=1e-10)
    close(sg, c["sigma"], rtol=1e-7, atol=1e-10)
    close(dmu, c["dmu"], rtol=1e-7, atol=1e-9 * np.abs(c["dmu"]).max())
    close(dsg, c["dsigma"], rtol=1e-6, atol=1e-8 * np.abs(c["dsigma"]).max())
    close(ei, c["ei"], rtol=1e-6, atol=1e-9 * np.abs(c["ei"]).max())
    close(dei, c["dei"], rtol=1e-6, atol=1e-8 * np.abs(c["dei"]).max())
    close(ucb, c["ucb"], rtol=1e-7, atol=1e-10)
    close(ducb, c["ducb"], rtol=1e-6, atol=1e-8 * np.abs(c["ducb"]).max())
    # the as-written (reference call structure) single-point forms on a few candidates, N <= 512 only (O(N^3) each)
    if X.shape[1] <= 512:
        for m in (0, 1, Xs.shape[1] - 1):
            x = Xs[:, m]
            close(r.predict_mu(x), c["mu"][m], rtol=1e-7, atol=1e-9)
            close(r.predict_sigma(x), c["sigma"][m], rtol=1e-6, atol=1e-9)
            close(r.predict_mu_derivative(x), c["dmu"][:, m], rtol=1e-6, atol=1e-8 * np.abs(c["dmu"]).max())
            close(r.predict_sigma_derivative(x), c["dsigma"][:, m], rtol=1e-5, atol=1e-7 * np.abs(c["dsigma"]).max())
            close(r.acq_value_as_written(x, oracle.ACQ_EI), c["ei"][m], rtol=1e-5, atol=1e-8 * np.abs(c["ei"]).max())
