"""Parity at BASELINE.json's full sizes.  Where the oracle finishes in seconds (C2) the comparison is direct; at C4 / C5
sizes the checks are size-independent identities of the GP posterior that hold for ANY correct
implementation and need no O(N^3) CPU work."""
import numpy as np
import pytest

from util import assert_starts_agree, sls, synth_candidates, synth_problem, synth_problem_with_signal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = sls().Context(0)
    yield c
    c.close()


def test_c2_gram_cholesky_predict_vs_oracle(ctx, oracle):
    """BASELINE config C2: N = 2048, D = 16, ARD-SE, 4096-point predict."""
    D, N, M = 16, 2048, 4096
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, M)
    ref = oracle.Regressor(X, y, theta, b, kernel=0)
    gp = sls().GP(ctx, X, y, theta, b, 0)
    mu_o, sg_o = ref.predict_batch(Xs)
    mu, sg = gp.predict(Xs)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(sg, sg_o, rtol=1e-6, atol=1e-8)
    ei_o, dei_o = ref.acq_eval_batch(Xs[:, :512])
    ei, dei = gp.acq_eval(Xs[:, :512])
    np.testing.assert_allclose(ei, ei_o, rtol=1e-6, atol=1e-9 * np.abs(ei_o).max())
    np.testing.assert_allclose(dei, dei_o, rtol=1e-6, atol=1e-7 * np.abs(dei_o).max())
    gp.close()


def _posterior_identities(gp, X, y, theta, b, n_check=384):
    """mu(x_i) = y_i - b alpha_i  and  sigma(x_i)^2 = b (1 - b (K^-1)_ii)  at the training points; K K^-1 = I on probes."""
    m = sls()
    N = X.shape[1]
    alpha = gp.matrix(m.GP_ALPHA)
    idx = np.linspace(0, N - 1, n_check).astype(int)
    mu, sg = gp.predict(X[:, idx])
    np.testing.assert_allclose(mu, (y - b * alpha)[idx], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(gp.matrix(m.GP_MU_DATA)[idx], (y - b * alpha)[idx], rtol=1e-12)
    Kinv = gp.matrix(m.GP_K_Y_INV)
    assert np.array_equal(Kinv, Kinv.T)
    np.testing.assert_allclose(sg ** 2, b * (1.0 - b * np.diag(Kinv)[idx]), rtol=1e-6, atol=1e-9)
    K = gp.matrix(m.GP_K_Y)
    rng = np.random.default_rng(0)
    V = rng.normal(size=(N, 4))
    R = K @ (Kinv @ V) - V
    assert np.abs(R).max() < 1e-7 * np.abs(V).max() * (theta[0] / b)
    np.testing.assert_allclose(K @ alpha, y, rtol=1e-6, atol=1e-7)
    L = gp.matrix(m.GP_CHOL_L)
    np.testing.assert_allclose((L @ (L.T @ V[:, :1]))[:, 0], (K @ V[:, :1])[:, 0], rtol=1e-9, atol=1e-10)
    assert abs(gp.summary()["logdet"] - 2.0 * np.log(np.diag(L)).sum()) < 1e-8 * N
    s = gp.summary()
    assert s["best_index"] == int(np.argmax(y - b * alpha))
    return alpha


@pytest.mark.parametrize("kernel", [0, 1])
def test_c4_size_fit_and_maximiser_properties(ctx, oracle, kernel):
    """BASELINE config C4 size: N = 8192, D = 64."""
    D, N, S = 64, 8192, 2048
    X, y, theta, b = synth_problem(oracle, D, N)
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    _posterior_identities(gp, X, y, theta, b)
    starts = synth_candidates(oracle, D, S)
    v0, g0 = gp.acq_eval(starts)
    assert np.all(v0 >= 0.0) and np.all(np.isfinite(g0))                 # EI >= 0
    # directional finite difference of EI along its own gradient, for the most promising starts
    top = np.argsort(-v0)[:8]
    for i in top:
        x = starts[:, i:i + 1]
        g = g0[:, i]
        if np.linalg.norm(g) == 0:
            continue
        d = g / np.linalg.norm(g)
        h = 1e-6
        fp = gp.acq_eval(np.clip(x + h * d[:, None], 0, 1), want_grad=False)[0]
        fm = gp.acq_eval(np.clip(x - h * d[:, None], 0, 1), want_grad=False)[0]
        np.testing.assert_allclose((fp - fm) / (2 * h), g @ d, rtol=2e-4, atol=1e-10)
    r = gp.acq_maximize(starts, 6)
    assert np.all(r["y_stars"] >= v0 - 1e-15)                             # monotone: never below the start
    assert r["value"] == r["y_stars"].max() and r["index"] == int(np.argmax(r["y_stars"]))
    assert np.all((r["x_stars"] >= 0) & (r["x_stars"] <= 1))
    np.testing.assert_allclose(gp.acq_eval(r["x"][:, None], want_grad=False)[0], r["value"], rtol=1e-7)
    # idempotence / determinism: the same call returns the same bits
    r2 = gp.acq_maximize(starts, 6)
    assert np.array_equal(r["y_stars"], r2["y_stars"]) and np.array_equal(r["x_stars"], r2["x_stars"])
    gp.close()


@pytest.mark.parametrize("kernel", [0, 1])
def test_c4_size_hip_vs_oracle(ctx, oracle, kernel):
    """BASELINE config C4 at its full N = 8192, D = 64: the HIP path against the CPU oracle (hoisted mode: blocked Cholesky,
    cached alpha and mu+) on identical inputs -- mu, sigma, EI and its gradient on 256 candidates, then the multi-start
    maximiser on 256 starts x 10 evaluations.  Tolerance 1e-6 relative (north_star).  The oracle fit is N^3 flops on the
    host cores (about 40 s on the 64-thread GPU box)."""
    D, N, M, S, n_local = 64, 8192, 256, 256, 10
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, M)
    ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    s = gp.summary()
    assert s["best_index"] == ref.best_index()          # hoisted form; the as-written O(N^3) loop agrees with it at N <= 2048
    np.testing.assert_allclose(s["mu_best"], ref.mu_best(), rtol=1e-7)
    mu_o, sg_o = ref.predict_batch(Xs)
    mu, sg = gp.predict(Xs)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(sg, sg_o, rtol=1e-6, atol=1e-9)
    dm_o, ds_o = ref.predict_grad_batch(Xs)
    dm, ds = gp.predict_grad(Xs)
    np.testing.assert_allclose(dm, dm_o, rtol=1e-6, atol=1e-8 * np.abs(dm_o).max())
    np.testing.assert_allclose(ds, ds_o, rtol=1e-6, atol=1e-7 * np.abs(ds_o).max())
    for acq, h in ((0, 1.0), (1, 2.0)):
        v_o, g_o = ref.acq_eval_batch(Xs, acq, h)
        v, g = gp.acq_eval(Xs, acq, h)
        np.testing.assert_allclose(v, v_o, rtol=1e-6, atol=1e-9 * np.abs(v_o).max())
        np.testing.assert_allclose(g, g_o, rtol=1e-6, atol=1e-7 * np.abs(g_o).max())
    starts = synth_candidates(oracle, D, S, seed=4321)
    ro = ref.acq_maximize(starts, n_local, diag=True)
    rg = gp.acq_maximize(starts, n_local)
    # at the headline size every divergent start must be EXPLAINED by a near-threshold Armijo test (no same-basin escape)
    assert_starts_agree(rg, ro, min_frac=0.97, label=f"C4 size kernel={kernel}", max_divergent=4)   # measured: 0 of 256
    np.testing.assert_allclose(rg["value"], ro["value"], rtol=1e-6)
    np.testing.assert_allclose(rg["x"], ro["x"], rtol=1e-6, atol=1e-7)
    assert ro["y_stars"][rg["index"]] >= ro["value"] * (1 - 1e-9)
    gp.close()


def test_c4_size_with_signal_hip_vs_oracle(ctx, oracle):
    """The C4 comparison once more at N = 8192, D = 64 on a target that carries signal at that dimension (util.
    synth_problem_with_signal): the posterior has structure, the EI maximiser is an INTERIOR point and the starts stay alive --
    the recipe's own target is below its noise at D = 64, so test_c4_size_hip_vs_oracle exercises starts that pin to the box.
    Same bar: mu, sigma, EI, gradients at 1e-6; every start that ends elsewhere must be explained by a near-threshold Armijo test
    (no same-basin escape); the chosen maximiser at 1e-6."""
    D, N, M, S, n_local = 64, 8192, 256, 256, 20
    X, y, theta, b = synth_problem_with_signal(oracle, D, N)
    assert y.std() > 0.03                                                     # well above the 0.01 noise level (0.0455 at D = 64)
    Xs = synth_candidates(oracle, D, M)
    ref = oracle.Regressor(X, y, theta, b, kernel=1)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    assert gp.summary()["best_index"] == ref.best_index()
    mu_o, sg_o = ref.predict_batch(Xs)
    mu, sg = gp.predict(Xs)
    assert mu_o.std() > 0.02                                                  # the posterior mean is not flat
    np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(sg, sg_o, rtol=1e-6, atol=1e-9)
    v_o, g_o = ref.acq_eval_batch(Xs)
    v, g = gp.acq_eval(Xs)
    np.testing.assert_allclose(v, v_o, rtol=1e-6, atol=1e-9 * np.abs(v_o).max())
    np.testing.assert_allclose(g, g_o, rtol=1e-6, atol=1e-7 * np.abs(g_o).max())
    # starts around the best data point (where EI lives), half a box width wide
    xb = X[:, ref.best_index()][:, None]
    starts = np.clip(xb + 0.5 * (synth_candidates(oracle, D, S, seed=4321) - 0.5), 0.0, 1.0)
    ro = ref.acq_maximize(starts, n_local, diag=True)
    rg = gp.acq_maximize(starts, n_local)
    st = gp.last_stats()
    interior = np.mean((rg["x"] > 1e-9) & (rg["x"] < 1 - 1e-9))
    assert interior > 0.5, interior                                           # most coordinates of the maximiser are not on the box
    assert st["evals_issued"] > 0.8 * S * n_local, st                         # the starts stay alive (the recipe's target: 69 %)
    assert_starts_agree(rg, ro, min_frac=0.97, label="C4 size with signal", max_divergent=6)
    np.testing.assert_allclose(rg["value"], ro["value"], rtol=1e-6)
    np.testing.assert_allclose(rg["x"], ro["x"], rtol=1e-6, atol=1e-7)
    gp.close()


def test_c5_size_map_objective_with_signal_vs_oracle(ctx, oracle):
    """C5 size (N = 4096, D = 128, Matern-5/2) on data with signal and ARD structure: objective and all 130 gradient components
    against the oracle at a genuinely anisotropic parameter vector (relevant coordinates short, irrelevant ones long)."""
    D, N = 128, 4096
    X, y, theta, b = synth_problem_with_signal(oracle, D, N, ard=True)
    h = sls().Nll(ctx, X, 1)
    ell = np.where(np.arange(D) % 4 == 0, 1.2, 8.0)
    x = np.concatenate([[0.3, 2e-4], ell])
    vo, go = oracle.gp_map_objective(1, X, y, x)
    v, g = h.gp_objective(y, x)
    np.testing.assert_allclose(v, vo, rtol=1e-9)
    np.testing.assert_allclose(g, go, rtol=1e-6, atol=1e-6 * np.abs(go).max())
    assert np.abs(go[2:][np.arange(D) % 4 == 0]).mean() > 5 * np.abs(go[2:][np.arange(D) % 4 != 0]).mean()   # the data speak about the relevant scales
    vb = h.gp_objective_batch(y, np.stack([x, x * 1.01]))
    np.testing.assert_allclose(vb[0], vo, rtol=1e-9)
    h.close()


def test_c5_size_map_objective_vs_oracle(ctx, oracle):
    """BASELINE config C5 at its full N = 4096, D = 128, Matern-5/2: MAP objective value and all D + 2 gradient components
    against the oracle's hoisted evaluation (Cholesky + fused W = alpha alpha^T - K^-1 contraction; its agreement with the
    reference's tensor + trace formulation is pinned at N = 200 in test_gpu_parity / test_oracle_golden), at theta_0 and
    at a perturbed, genuinely ARD theta."""
    D, N = 128, 4096
    X, y, theta, b = synth_problem(oracle, D, N)
    h = sls().Nll(ctx, X, 1)
    rng = np.random.default_rng(1237)
    for x in (np.concatenate([[0.5, 0.005], np.full(D, theta[1])]),
              np.concatenate([[0.7, 0.02], theta[1] * rng.uniform(0.7, 1.4, D)])):
        vo, go = oracle.gp_map_objective(1, X, y, x)
        v, g = h.gp_objective(y, x)
        np.testing.assert_allclose(v, vo, rtol=1e-9)
        np.testing.assert_allclose(g, go, rtol=1e-6, atol=1e-6 * np.abs(go).max())
    h.close()


def test_c5_size_map_gradient(ctx, oracle):
    """BASELINE config C5 size: Matern-5/2 MAP objective + gradient at N = 4096, D = 128.
    The gradient is checked against central differences of the device objective itself (value parity with the oracle
    is covered at N <= 200) and the noise-gradient against its closed form 1/2 (alpha.alpha - tr K^-1) + prior'."""
    D, N = 128, 4096
    X, y, theta, b = synth_problem(oracle, D, N)
    h = sls().Nll(ctx, X, 1)
    x = np.concatenate([[0.5, 0.005], np.full(D, theta[1])])
    v, g = h.gp_objective(y, x)
    assert np.isfinite(v) and np.all(np.isfinite(g))
    for p in (0, 1, 2, 2 + D // 2, 1 + D):
        e = np.zeros_like(x)
        e[p] = 1e-5 * x[p]
        fd = (h.gp_objective(y, x + e, want_grad=False) - h.gp_objective(y, x - e, want_grad=False)) / (2 * e[p])
        np.testing.assert_allclose(g[p], fd, rtol=2e-4, atol=1e-4)
    r = h.eval(y, np.concatenate([[x[0]], x[2:]]), x[1])
    gp = sls().GP(ctx, X, y, np.concatenate([[x[0]], x[2:]]), x[1], 1)
    Kinv = gp.matrix(sls().GP_K_Y_INV)
    np.testing.assert_allclose(r["grad_b"], 0.5 * (r["alpha"] @ r["alpha"] - np.trace(Kinv)), rtol=1e-7)
    np.testing.assert_allclose(r["quad"], y @ r["alpha"], rtol=1e-10)
    gp.close()
    h.close()


@pytest.mark.parametrize("kernel", [0, 1])
def test_acq_gemm_scheduling_variants_agree(ctx, oracle, kernel, monkeypatch):
    """The tile-scheduling variants of acq_gemm_kernel (one or two workgroups per CU; one tile per workgroup or the persistent
    generation-gated form with one or two gate groups per XCD) only change the ORDER in which tiles run: values and gradients
    must be bit-identical across them, and agree with the oracle, at a size where the gated forms are active (N = 2048 -> 16
    row tiles, 8192 candidates -> 64 column tiles = 1024 tiles = two to four generations)."""
    D, N, M = 16, 2048, 8192
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, M)
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    ctx.set_candidate_chunk(16384)
    monkeypatch.setenv("SLS_WAVE_PATH", "0")
    base = None
    for env in ({"SLS_ACQ_WG_PER_CU": "1", "SLS_PERSIST": "1", "SLS_GATE_PHASE": "0", "SLS_GATE_EVERY": "16"},   # the default (round 5)
                {"SLS_ACQ_WG_PER_CU": "1", "SLS_PERSIST": "1", "SLS_GATE_PHASE": "0", "SLS_GATE_EVERY": "1"},
                {"SLS_ACQ_WG_PER_CU": "1", "SLS_PERSIST": "2", "SLS_GATE_PHASE": "2000"},                          # ungated (rounds 3-4)
                {"SLS_ACQ_WG_PER_CU": "1", "SLS_PERSIST": "0", "SLS_GATE_PHASE": "2000"},
                {"SLS_ACQ_WG_PER_CU": "1", "SLS_PERSIST": "1", "SLS_GATE_PHASE": "2000"},
                {"SLS_ACQ_WG_PER_CU": "2", "SLS_PERSIST": "0", "SLS_GATE_PHASE": "2000"},
                {"SLS_ACQ_WG_PER_CU": "2", "SLS_PERSIST": "1", "SLS_GATE_PHASE": "2000"},
                {"SLS_ACQ_WG_PER_CU": "2", "SLS_PERSIST": "1", "SLS_GATE_PHASE": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        val, grad = gp.acq_eval(Xs)
        mu, sg = gp.predict(Xs)
        if base is None:
            base = (val, grad, mu, sg)
            ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
            v_o, g_o = ref.acq_eval_batch(Xs[:, :256])
            np.testing.assert_allclose(val[:256], v_o, rtol=1e-6, atol=1e-9 * np.abs(v_o).max())
            np.testing.assert_allclose(grad[:, :256], g_o, rtol=1e-6, atol=1e-7 * np.abs(g_o).max())
            # gradient-free prediction: triangular var_gemm, here on its XCD-grouped long-first tile order (1024 tiles)
            mu_o, sg_o = ref.predict_batch(Xs[:, -256:])
            np.testing.assert_allclose(mu[-256:], mu_o, rtol=1e-6, atol=1e-8)
            np.testing.assert_allclose(sg[-256:], sg_o, rtol=1e-6, atol=1e-8)
        else:
            # the k loop of a tile does not depend on when or where the tile runs
            np.testing.assert_array_equal(mu, base[2])
            np.testing.assert_array_equal(sg, base[3])
            np.testing.assert_array_equal(val, base[0])
            np.testing.assert_array_equal(grad, base[1])
    gp.close()


@pytest.mark.parametrize("M", [100, 4736, 8960])
def test_acq_gemm_tail_split_is_bit_identical(ctx, oracle, M, monkeypatch):
    """When the last 512-tile generation of an acq_gemm launch is at most half full its tiles run as HALF tiles on twice as
    many workgroups (SLS_TAIL_SPLIT, default on).  Values and gradients must be bit-identical to the unsplit schedule: M = 100
    (16 tiles, all split), 4736 (592 tiles, one-tile-per-workgroup form, 80 in the tail), 8960 (1120 tiles, persistent gated
    form, 96 in the tail), N = 2048."""
    D, N = 16, 2048
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, M)
    monkeypatch.setenv("SLS_WAVE_PATH", "0")
    gp = sls().GP(ctx, X, y, theta, b, 1)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SLS_TAIL_SPLIT", flag)
        out[flag] = gp.acq_eval(Xs) + gp.predict_grad(Xs)
    for a_, b_ in zip(out["1"], out["0"]):
        assert np.array_equal(a_, b_)
    ref = oracle.Regressor(X, y, theta, b, kernel=1)
    v_o, g_o = ref.acq_eval_batch(Xs[:, -64:])
    np.testing.assert_allclose(out["1"][0][-64:], v_o, rtol=1e-6, atol=1e-9 * np.abs(v_o).max())
    np.testing.assert_allclose(out["1"][1][:, -64:], g_o, rtol=1e-6, atol=1e-7 * np.abs(g_o).max())
    gp.close()


def test_beyond_headline_size_n16384(ctx, oracle):
    """Twice the headline N (N = 16 384: 2 GB per N x N matrix, 128 block steps of the Cholesky, 7 trtri levels): the
    size-independent posterior identities must still hold; the predictions here take the triangular var_gemm path on its
    XCD-grouped tile order (128 x 8 tile groups) and the value+gradient path the gated acq_gemm with 16 generations."""
    D, N = 32, 16384
    X, y, theta, b = synth_problem(oracle, D, N)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    _posterior_identities(gp, X, y, theta, b, n_check=1024)
    Xs = synth_candidates(oracle, D, 2048)
    v, g = gp.acq_eval(Xs)
    v0 = gp.acq_eval(Xs, want_grad=False)
    np.testing.assert_allclose(v0, v, rtol=1e-6, atol=1e-9 * np.abs(v).max())
    h = 1e-6
    d = np.random.default_rng(3).normal(size=(D, 1)); d /= np.linalg.norm(d)
    pick = np.argsort(-v)[:8]
    fp = gp.acq_eval(Xs[:, pick] + h * d, want_grad=False)
    fm = gp.acq_eval(Xs[:, pick] - h * d, want_grad=False)
    np.testing.assert_allclose((fp - fm) / (2 * h), (g[:, pick] * d).sum(axis=0), rtol=2e-4, atol=1e-7 * np.abs(g).max())
    gp.close()


def test_c4_full_workload_once(ctx, oracle, monkeypatch):
    """BASELINE config C4 at its FULL size, once: N = 8192, D = 64, 65 536 starts, cap 50 evaluations per start (what bench.py
    times).  A start's arithmetic does not depend on the columns it shares a tile with nor on how many starts are still
    live, so the first 4 096 starts of the full run must end bit for bit where the same 4 096 starts end on their own without
    the active-set compaction (SLS_COMPACT=0: every start evaluated every round); the winner of the full run must be the
    first maximum of its own end values."""
    from util import record
    D, N, S, n_local, SUB = 64, 8192, 65536, 50, 4096
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S, seed=1236)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    import time
    t0 = time.time()
    full = gp.acq_maximize(starts, n_local)
    t_full = time.time() - t0
    st = gp.last_stats()
    assert full["index"] == int(np.argmax(full["y_stars"]))               # first maximum (Eigen maxCoeff)
    assert full["value"] == full["y_stars"][full["index"]]
    np.testing.assert_array_equal(full["x"], full["x_stars"][:, full["index"]])
    monkeypatch.setenv("SLS_COMPACT", "0")
    sub = gp.acq_maximize(np.asfortranarray(starts[:, :SUB]), n_local)
    assert np.array_equal(sub["y_stars"], full["y_stars"][:SUB])
    assert np.array_equal(sub["x_stars"], full["x_stars"][:, :SUB])
    record("c4_full_workload", seconds_wall=t_full, stats=st, value=float(full["value"]), index=int(full["index"]),
           finite=int(np.isfinite(full["y_stars"]).sum()))
    gp.close()
