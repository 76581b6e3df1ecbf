"""The run-time switches (csrc/tuning.hpp) that no other test file turns: each must leave the results where they were -- identical
bits where only the data's route changes, rounding where the arithmetic's order does.  (The schedule switches of the Cholesky
kernels are in test_gpu_parity.py; this file closes the list that DESIGN.md section 9 gives.)"""
import numpy as np
import pytest

from util import sls, synth_candidates, synth_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,D,kernel", [(60, 5, 1), (128, 17, 0)])
def test_small_fit_kernel_against_the_tiled_pipeline(oracle, N, D, kernel, monkeypatch):
    """SLS_FIT_SMALL: N <= 128 fits in ONE single-workgroup launch (gp_fit_small_kernel) or through the tiled pipeline (Gram + potri +
    gemv): two orders of the same sums (src/gaussian-process-regressor.cpp:198-232)."""
    m = sls()
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, 40)
    out = {}
    for name, v in (("small", "1"), ("tiled", "0")):
        monkeypatch.setenv("SLS_FIT_SMALL", v)
        c = m.Context(0)
        g = m.GP(c, X, y, theta, b, kernel)
        out[name] = (g.matrix(m.GP_ALPHA), g.matrix(m.GP_K_Y_INV), *g.predict(Xs), g.summary()["logdet"])
        g.close(); c.close()
    monkeypatch.delenv("SLS_FIT_SMALL")
    ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
    mu_o, sg_o = ref.predict_batch(Xs)
    for name, (alpha, Kinv, mu, sg, logdet) in out.items():
        np.testing.assert_allclose(mu, mu_o, rtol=1e-8, atol=1e-10, err_msg=name)
        np.testing.assert_allclose(sg, sg_o, rtol=1e-6, atol=1e-9, err_msg=name)
    a_s, K_s, mu_s, sg_s, ld_s = out["small"]
    a_t, K_t, mu_t, sg_t, ld_t = out["tiled"]
    np.testing.assert_allclose(a_s, a_t, rtol=1e-8, atol=1e-10 * np.abs(a_t).max())
    np.testing.assert_allclose(K_s, K_t, rtol=1e-8, atol=1e-10 * np.abs(K_t).max())
    assert abs(ld_s - ld_t) <= 1e-10 * max(1.0, abs(ld_t))


def test_small_objective_results_through_mapped_memory_or_a_copy(oracle, monkeypatch):
    """SLS_SMALL_ZEROCOPY: the small MAP objective writes its results into a mapped host block, or into device memory that is copied
    back: the same kernel, the same bits (src/gaussian-process-regressor.cpp:36-193)."""
    m = sls()
    D, N = 7, 90
    X, y, theta, b = synth_problem(oracle, D, N)
    x = np.concatenate([[0.7, 0.02], np.linspace(0.3, 0.8, D)])
    res = {}
    for v in ("1", "0"):
        monkeypatch.setenv("SLS_SMALL_ZEROCOPY", v)
        c = m.Context(0)
        h = m.Nll(c, X, 1)
        res[v] = [h.gp_objective(y, x), h.gp_objective(y * 1.5, x), h.gp_objective(y, x * 1.01)]
        h.close(); c.close()
    monkeypatch.delenv("SLS_SMALL_ZEROCOPY")
    for (v1, g1), (v0, g0) in zip(res["1"], res["0"]):
        assert v1 == v0 and np.array_equal(g1, g0)
    vo, go = oracle.gp_map_objective(1, X, y, x)
    assert abs(res["1"][0][0] - vo) <= 1e-9 * abs(vo)
    np.testing.assert_allclose(res["1"][0][1], go, rtol=1e-6, atol=1e-8 * np.abs(go).max())


def test_one_process_many_devices_with_and_without_rccl(oracle, monkeypatch):
    """SLS_MULTI_RCCL: sls_multi merges the shards' winners with ONE ncclAllGather, or on the host (the form it also takes when a device
    is listed twice): the same winner, value and point (src/acquisition-function.cpp:121-153: maxCoeff over the starts)."""
    m = sls()
    D, N, S = 4, 200, 96
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    res = {}
    for v in ("1", "0"):
        monkeypatch.setenv("SLS_MULTI_RCCL", v)
        mu = m.Multi([0])
        note = mu.exchange
        g = m.MultiGP(mu, X, y, theta, b, 1)
        r = g.acq_maximize(starts, 12)
        res[v] = (note, r)
        g.close(); mu.close()
    monkeypatch.delenv("SLS_MULTI_RCCL")
    assert "SLS_MULTI_RCCL=0" in res["0"][0]
    assert res["1"][0].startswith("ncclAllGather") or res["1"][0].startswith("host merge")     # RCCL absent: says why
    r1, r0 = res["1"][1], res["0"][1]
    assert r1["index"] == r0["index"] and r1["value"] == r0["value"] and np.array_equal(r1["x"], r0["x"])


def test_device_memory_pool_limit(oracle, monkeypatch):
    """SLS_POOL_MB: freed device buffers are kept for reuse up to this many megabytes; with 0 every buffer goes back to the driver.  The
    same results either way."""
    m = sls()
    D, N = 5, 700
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, 50)
    out = {}
    for v in ("16384", "0"):
        monkeypatch.setenv("SLS_POOL_MB", v)
        c = m.Context(0)
        r = []
        for _ in range(3):
            g = m.GP(c, X, y, theta, b, 1)
            r.append(g.predict(Xs))
            g.close()
        out[v] = r
        c.close()
    monkeypatch.delenv("SLS_POOL_MB")
    for (m1, s1), (m0, s0) in zip(out["16384"], out["0"]):
        assert np.array_equal(m1, m0) and np.array_equal(s1, s0)
    assert np.array_equal(out["0"][0][0], out["0"][2][0])
