"""GPU parity: the HIP path (through the C ABI, include/sls_hip.h) against the CPU oracle on identical inputs.

Tolerance: BASELINE.json's north_star asks for 1e-6 relative in fp64; the assertions below use 1e-6 or tighter
(absolute floors only where the quantity itself underflows / cancels to ~0)."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from util import assert_starts_agree, env_switch, oracle_end_value_sensitivity, record, relerr, sls, synth_candidates, synth_problem

pytestmark = pytest.mark.gpu

RTOL = 1e-6


@pytest.fixture(scope="module")
def ctx():
    c = sls().Context(0)
    yield c
    c.close()


@pytest.fixture(params=["wave", "tiled"])
def path(request, monkeypatch):
    """Small problems take the per-point wavefront kernels (kernels_wave.hip) by default; SLS_WAVE_PATH=0 forces the tiled
    MFMA pipeline that large problems use.  Tests that request this fixture run on both."""
    monkeypatch.setenv("SLS_WAVE_PATH", "1" if request.param == "wave" else "0")
    return request.param


def close(a, b, rtol=RTOL, atol=0.0):
    np.testing.assert_allclose(np.asarray(a, dtype=float), np.asarray(b, dtype=float), rtol=rtol, atol=atol)


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("D,N", [(1, 7), (5, 128), (16, 300), (33, 129)])
def test_gram(ctx, oracle, kernel, D, N):
    X, y, theta, b = synth_problem(oracle, D, N)
    theta[1:] *= np.linspace(0.7, 1.3, D)          # genuinely ARD
    K = ctx.gram(X, theta, b, kernel)
    Ko = oracle.calc_large_ky(kernel, X, theta, b)
    close(K, Ko, rtol=1e-10, atol=1e-14)
    assert np.array_equal(K, K.T)
    np.testing.assert_allclose(np.diag(K), theta[0] + b, rtol=1e-15)
    Xs = synth_candidates(oracle, D, 37)
    Ks = ctx.gram_cross(X, Xs, theta, kernel)
    Kso = np.stack([oracle.calc_small_k(kernel, Xs[:, m], X, theta) for m in range(Xs.shape[1])], axis=1)
    close(Ks, Kso, rtol=1e-10, atol=1e-14)


@pytest.mark.parametrize("N", [1, 50, 128, 300, 640])
def test_cholesky_solve_inverse(ctx, oracle, N):
    rng = np.random.default_rng(N)
    A = rng.normal(size=(N, N))
    A = A @ A.T + N * np.eye(N)
    L = ctx.potrf(A)
    Lo, info = oracle.cholesky(A)
    assert info == 0
    close(L, Lo, rtol=1e-10, atol=1e-12)
    assert np.all(np.triu(L, 1) == 0)
    B = rng.normal(size=(N, 3))
    close(ctx.potrs(L, B), oracle.chol_solve(Lo, B), rtol=1e-9, atol=1e-12)
    close(ctx.potrs(L, B[:, 0]), oracle.chol_solve(Lo, B[:, 0]), rtol=1e-9, atol=1e-12)
    Ai = ctx.potri(L)
    close(Ai, oracle.spd_inverse_from_chol(Lo), rtol=1e-8, atol=1e-12)
    assert np.array_equal(Ai, Ai.T)


@pytest.mark.parametrize("N", [384, 1000, 2304])
def test_potrf_schedules_agree(N, monkeypatch):
    """The Cholesky schedules (kernels_chol.hip): the single-launch dataflow kernel (default: per-tile ownership + ready flags,
    panel tiles by triangular solves; single steps and chunked updates) and the multi-launch forms
    (SLS_POTRF_MODE=0, the fallback: one-level, two-level, two-level with the outer update on a CU-masked side stream).  Who
    computes a tile never changes what is computed: variants that differ only in the schedule give identical bits; the dataflow
    form solves its panel tiles against L_jj where the multi-launch form multiplies by T_jj, and chunked updates sum the outer
    update in one k loop: agreement to rounding there.  Every variant must also reject an indefinite matrix."""
    rng = np.random.default_rng(N)
    B = rng.normal(size=(N, N))
    A = B @ B.T / N + np.eye(N)
    res = {}
    for name, env in (("multi", {"SLS_POTRF_MODE": "0", "SLS_POTRF_NBO": "1"}),
                      ("multi2", {"SLS_POTRF_MODE": "0", "SLS_POTRF_NBO": "2"}),
                      ("dataflow1", {"SLS_POTRF_MODE": "3", "SLS_POTRF_DNBO": "1"}),
                      # the round-3 chain (solve of the panel tile (j+1, j) on the chain itself); the streamed form without the
                      # half-tile owners, with them for the sub-diagonal tiles only, and for a band of three
                      ("dataflow1_nostream", {"SLS_POTRF_MODE": "3", "SLS_POTRF_DNBO": "1", "SLS_POTRF_STREAM": "0"}),
                      ("dataflow1_nosplit", {"SLS_POTRF_MODE": "3", "SLS_POTRF_DNBO": "1", "SLS_POTRF_SPLIT": "0"}),
                      ("dataflow1_band1", {"SLS_POTRF_MODE": "3", "SLS_POTRF_DNBO": "1", "SLS_POTRF_SPLIT": "1"}),
                      ("dataflow1_band3", {"SLS_POTRF_MODE": "3", "SLS_POTRF_DNBO": "1", "SLS_POTRF_SPLIT": "3"}),
                      # round 6: the chain as three workgroups in rotation / two taking turns; the workers' tiles claimed from pools per XCD
                      ("dataflow1_chain3", {"SLS_POTRF_MODE": "3", "SLS_POTRF_DNBO": "1", "SLS_POTRF_FUSE_SYRK": "1"}),
                      ("dataflow1_chain2", {"SLS_POTRF_MODE": "3", "SLS_POTRF_DNBO": "1", "SLS_POTRF_FUSE_SYRK": "0"}),
                      ("dataflow1_pool", {"SLS_POTRF_MODE": "3", "SLS_POTRF_DNBO": "1", "SLS_POTRF_POOL": "1"}),
                      ("dataflow2", {"SLS_POTRF_MODE": "3", "SLS_POTRF_DNBO": "2"}),
                      ("dataflow4near", {"SLS_POTRF_MODE": "3", "SLS_POTRF_DNBO": "4", "SLS_POTRF_DNEAR": "2"})):
        for k in [k for k in os.environ if k.startswith("SLS_POTRF_")]:
            monkeypatch.delenv(k)      # every variant starts from the defaults
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = sls().Context(0)
        res[name] = c.potrf(A)
        bad = A.copy(); bad[N - 5, N - 5] = -1.0
        with pytest.raises(sls().SlsError):
            c.potrf(bad)
        assert c.prof_get("potrf_fallbacks")[1] == 0, name
        c.close()
    L = np.linalg.cholesky(A)
    for name, v in res.items():
        close(v, L, rtol=1e-10, atol=1e-12)
    # panel tiles by triangular solves against L_jj (16 x 16 inverses) instead of products with T_jj: rounding
    close(res["dataflow1"], res["multi"], rtol=1e-12, atol=1e-13)
    close(res["dataflow2"], res["multi"], rtol=1e-12, atol=1e-13)
    close(res["dataflow4near"], res["multi"], rtol=1e-12, atol=1e-13)
    # ... and so do the follower workgroup, the streamed solves and the half-tile owners (same slabs, same MFMA order)
    for name in ("dataflow1_nostream", "dataflow1_nosplit", "dataflow1_band1", "dataflow1_band3", "dataflow1_chain3", "dataflow1_chain2",
                 "dataflow1_pool"):
        assert np.array_equal(res[name], res["dataflow1"]), name
    close(res["multi2"], res["multi"], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("N,D", [(300, 4), (640, 7), (1500, 16), (2304, 5)])
def test_fused_inverse_matches_separate_launches(oracle, N, D, monkeypatch):
    """N <= 4096: ONE launch factors K_y and builds L^-1, its transpose and K_y^-1 behind the factorisation's chain (kernels_chol.hip:
    potri_team), in place of potrf + trtri + lauum (src/gaussian-process-regressor.cpp:159, 211, 231: MatrixXd::inverse()).  The
    factor must have the bits of the factorisation alone (who shares the chip never changes a tile's arithmetic), the inverse
    must agree with the separate launches to rounding (block forward substitution vs recursive doubling) and with LAPACK, the result
    must not depend on how the chip is split between the two teams, and a forced give-up of the launch must end in the
    multi-launch recomputation."""
    m = sls()
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, 96)
    out = {}
    knobs = ("SLS_POTRI_FUSED", "SLS_POTRI_W1", "SLS_POTRI_CX", "SLS_POTRI_CK", "SLS_POTRI_PLAST")
    for name, env in (("fused", {}), ("separate", {"SLS_POTRI_FUSED": "0"}), ("fused_small_team", {"SLS_POTRI_W1": "7"}),
                      ("fused_chunks", {"SLS_POTRI_CX": "3", "SLS_POTRI_CK": "2"}),
                      # the last term of every row split off (one product per row on the column wavefront) / not
                      ("fused_plast0", {"SLS_POTRI_PLAST": "0"}), ("fused_plast1", {"SLS_POTRI_PLAST": "1"})):
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = m.Context(0)
        c.prof_enable(True)
        g = m.GP(c, X, y, theta, b, 1)
        mu, sg = g.predict(Xs)
        out[name] = dict(L=g.matrix(m.GP_CHOL_L), Kinv=g.matrix(m.GP_K_Y_INV), alpha=g.matrix(m.GP_ALPHA), mu=mu, sg=sg,
                         K=g.matrix(m.GP_K_Y), launches=c.prof_get("potri")[1], logdet=g.summary()["logdet"])
        assert c.prof_get("potrf_fallbacks")[1] == 0
        g.close(); c.close()
    f, s = out["fused"], out["separate"]
    assert f["launches"] == 1 and s["launches"] == 0                       # the fused launch really ran / really did not
    assert np.array_equal(f["L"], s["L"]) and f["logdet"] == s["logdet"]
    scale = np.abs(s["Kinv"]).max()
    close(f["Kinv"], s["Kinv"], rtol=1e-9, atol=1e-11 * scale)
    assert np.array_equal(f["Kinv"], f["Kinv"].T)
    close(f["alpha"], s["alpha"], rtol=1e-8, atol=1e-10 * np.abs(s["alpha"]).max())
    close(f["mu"], s["mu"], rtol=1e-9, atol=1e-12)
    close(f["sg"], s["sg"], rtol=1e-7, atol=1e-12)
    # against LAPACK: K_y K_y^-1 = I
    R = f["K"] @ f["Kinv"] - np.eye(N)
    assert np.abs(R).max() < 1e-9 * np.linalg.cond(f["K"]) ** 0.5 + 1e-10, np.abs(R).max()
    close(f["Kinv"], np.linalg.inv(f["K"]), rtol=0, atol=1e-9 * scale)
    # another split of the chip: same bits; another (fixed) chunking of the accumulations: rounding only
    for key in ("L", "Kinv", "alpha", "mu", "sg"):
        assert np.array_equal(out["fused_small_team"][key], f[key]), key
    for name in ("fused_chunks", "fused_plast0", "fused_plast1"):
        close(out[name]["Kinv"], f["Kinv"], rtol=1e-9, atol=1e-11 * scale)
        assert np.array_equal(out[name]["L"], f["L"])
        assert np.array_equal(out[name]["Kinv"], out[name]["Kinv"].T)
    # forced expiry of the device-side waits: the fit is recomputed with separate launches
    for k in knobs:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SLS_POTRF_TIMEOUT_TICKS", "1")
    c = m.Context(0)
    g = m.GP(c, X, y, theta, b, 1)
    Kg = g.matrix(m.GP_K_Y_INV)
    assert c.prof_get("potrf_fallbacks")[1] == 1
    close(Kg, s["Kinv"], rtol=1e-9, atol=1e-11 * scale)
    g.close(); c.close()


@pytest.mark.parametrize("N,D", [(640, 3), (1500, 5), (2500, 6), (3000, 9), (3600, 4)])
def test_pools_chain_forms_and_static_teams_give_identical_bits(oracle, N, D, monkeypatch):
    """Round 6's schedules of the fused factor + inverse (kernels_chol.hip): dynamic pools per XCD (default from N = 2432), with the
    diagonal tiles owned statically beside them (default up to N = 3456) or pooled too, with and without keeping an item across
    tasks; static teams (SLS_POTRI_POOL=0); the chain as three workgroups in rotation (default at N = 1536-2560) or two taking
    turns.  Who runs a task, and when, never changes what it computes: the factor, K^-1 and alpha of every variant carry the bits of
    the default (src/gaussian-process-regressor.cpp:159,211: one MatrixXd::inverse() in the reference)."""
    m = sls()
    X, y, theta, b = synth_problem(oracle, D, N)
    knobs = ("SLS_POTRI_POOL", "SLS_POTRI_POOL_KEEP", "SLS_POTRI_POOL_NEAR", "SLS_POTRI_POOL_NEAR_W", "SLS_POTRF_FUSE_SYRK", "SLS_POTRI_W1")
    out = {}
    for name, env in (("default", {}), ("static", {"SLS_POTRI_POOL": "0"}), ("pool_nokeep", {"SLS_POTRI_POOL": "1", "SLS_POTRI_POOL_KEEP": "0"}),
                      ("pool_all", {"SLS_POTRI_POOL": "1", "SLS_POTRI_POOL_NEAR": "-1"}),
                      ("pool_near1", {"SLS_POTRI_POOL": "1", "SLS_POTRI_POOL_NEAR": "1", "SLS_POTRI_POOL_NEAR_W": "16"}),
                      # a band too wide for ONE owner's table (the launcher then pools everything) / a wide band that fits two
                      ("pool_near_wide1", {"SLS_POTRI_POOL": "1", "SLS_POTRI_POOL_NEAR": "8", "SLS_POTRI_POOL_NEAR_W": "1"}),
                      ("pool_near_wide4", {"SLS_POTRI_POOL": "1", "SLS_POTRI_POOL_NEAR": "6", "SLS_POTRI_POOL_NEAR_W": "4"}),
                      # fewer workgroups in the factorisation's team than owners of the diagonal tiles: some owners sit in the other team
                      ("pool_small_team", {"SLS_POTRI_POOL": "1", "SLS_POTRI_W1": "7"}),
                      ("pool_big_team", {"SLS_POTRI_POOL": "1", "SLS_POTRI_W1": "200"}),
                      ("chain3", {"SLS_POTRF_FUSE_SYRK": "1"}), ("chain2", {"SLS_POTRF_FUSE_SYRK": "0"}),
                      ("chain3_static", {"SLS_POTRF_FUSE_SYRK": "1", "SLS_POTRI_POOL": "0", "SLS_POTRI_W1": "90"})):
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = m.Context(0)
        c.prof_enable(True)
        g = m.GP(c, X, y, theta, b, 1)
        out[name] = dict(L=g.matrix(m.GP_CHOL_L), Kinv=g.matrix(m.GP_K_Y_INV), alpha=g.matrix(m.GP_ALPHA))
        assert c.prof_get("potri")[1] == 1 and c.prof_get("potrf_fallbacks")[1] == 0, name     # ONE fused launch, which did not give up
        g.close(); c.close()
    ref = out["default"]
    assert np.array_equal(ref["Kinv"], ref["Kinv"].T)
    for name, v in out.items():
        for key in ("L", "Kinv", "alpha"):
            assert np.array_equal(v[key], ref[key]), (name, key, float(np.abs(v[key] - ref[key]).max()))
    Lr = ref["L"]
    A = Lr @ Lr.T
    R = A @ ref["Kinv"] - np.eye(N)
    assert np.abs(R).max() < 1e-9 * np.linalg.cond(A) ** 0.5 + 1e-10, np.abs(R).max()
    # forced expiry of every device-side wait (pool scans included): the launch gives up, the fit is recomputed with separate launches
    for k in knobs:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SLS_POTRF_TIMEOUT_TICKS", "1")
    c = m.Context(0)
    c.prof_enable(True)
    g = m.GP(c, X, y, theta, b, 1)
    Kg = g.matrix(m.GP_K_Y_INV)
    assert c.prof_get("potrf_fallbacks")[1] == 1
    close(Kg, ref["Kinv"], rtol=1e-9, atol=1e-11 * np.abs(ref["Kinv"]).max())
    g.close(); c.close()
    monkeypatch.delenv("SLS_POTRF_TIMEOUT_TICKS")


@pytest.mark.parametrize("N", [640, 1500, 2500, 3000])
def test_singular_matrix_is_rejected_by_every_fused_form(oracle, N):
    """A fit whose K_y is singular (exact duplicates among the data points, no noise) must END -- with the error the multi-launch
    form reports, at once, without the give-up path -- in every form of the single launch: static teams, the three-workgroup
    chain (N = 1500, 2500), dynamic pools (2500, 3000).  The chain that meets the pivot raises the abort flag; every workgroup
    of both teams and every pool's scan polls it.  Measured: 0.4-2.1 ms from call to exception; the context stays usable.
    (The reference asserts nothing here: Eigen's LLT of a singular K silently yields NaNs, src/preference-regressor.cpp:293-330.)"""
    import re
    m = sls()
    X, y, theta, b = synth_problem(oracle, 6, N)
    c = m.Context(0)
    c.prof_enable(True)
    m.GP(c, X, y, theta, b, 1).close()
    for dup_from in (N - 3, N // 2):
        Xb = X.copy()
        Xb[:, dup_from:] = X[:, :N - dup_from]
        t0 = time.perf_counter()
        with pytest.raises(m.SlsError, match="not positive definite") as e:
            m.GP(c, Xb, y, theta, 0.0, 1)
        dt = time.perf_counter() - t0
        pivot = int(re.search(r"pivot (\d+)", str(e.value)).group(1))
        assert dup_from <= pivot <= N, (pivot, dup_from)          # the first duplicate's row, or a later one if rounding left it a positive pivot
        assert dt < 0.1, dt                                       # the give-up path alone would take 0.2 s
        assert c.prof_get("potrf_fallbacks")[1] == 0
    g = m.GP(c, X, y, theta, b, 1)
    mu, _ = g.predict(X[:, :8])
    assert np.abs(mu - y[:8]).max() < 0.2
    g.close(); c.close()


@pytest.mark.parametrize("N,D", [(2304, 5)])
def test_separate_launch_inverse_switches_give_identical_bits(oracle, N, D, monkeypatch):
    """potrf + trtri + lauum (the form of N > 4096, forced here with SLS_POTRI_FUSED=0): half or whole tiles per level of the
    recursive doubling (SLS_TRTRI_NARROW), one or two workgroups per CU (SLS_TRI_WG_PER_CU), half-tile lauum (SLS_LAUUM_N64): same slabs,
    same fragments, same k order -- identical bits (kernels_tri.hip)."""
    m = sls()
    X, y, theta, b = synth_problem(oracle, D, N)
    knobs = ("SLS_POTRI_FUSED", "SLS_TRTRI_NARROW", "SLS_TRI_WG_PER_CU", "SLS_LAUUM_N64")
    out = {}
    for name, env in (("default", {}), ("all_whole", {"SLS_TRTRI_NARROW": "0", "SLS_LAUUM_N64": "0"}),
                      ("all_half", {"SLS_TRTRI_NARROW": "100000", "SLS_LAUUM_N64": "1"}), ("wg1", {"SLS_TRI_WG_PER_CU": "1"}),
                      ("wg2", {"SLS_TRI_WG_PER_CU": "2"})):
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("SLS_POTRI_FUSED", "0")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = m.Context(0)
        c.prof_enable(True)
        g = m.GP(c, X, y, theta, b, 1)
        out[name] = dict(L=g.matrix(m.GP_CHOL_L), Kinv=g.matrix(m.GP_K_Y_INV), alpha=g.matrix(m.GP_ALPHA))
        assert c.prof_get("potri")[1] == 0, name
        g.close(); c.close()
    for name, v in out.items():
        for key in ("L", "Kinv", "alpha"):
            assert np.array_equal(v[key], out["default"][key]), (name, key)


@pytest.mark.parametrize("N,D", [(700, 5), (2100, 12), (2700, 9), (3500, 4)])      # the last two: dynamic pools (with / without owned diagonal tiles)
def test_fused_inverse_inside_the_map_objective(oracle, N, D, monkeypatch):
    """The GP MAP objective + gradient (sls_gp_nll_grad, src/gaussian-process-regressor.cpp:36-193) on the fused factor + inverse
    launch against the same evaluation on separate launches; with the pools switched off: the same bits."""
    m = sls()
    X, y, _, _ = synth_problem(oracle, D, N)
    x = np.concatenate([[0.6, 0.01], np.linspace(0.4, 0.9, D)])
    res = {}
    for name, env in (("fused", {}), ("separate", {"SLS_POTRI_FUSED": "0"}), ("fused_static", {"SLS_POTRI_POOL": "0"})):
        monkeypatch.delenv("SLS_POTRI_FUSED", raising=False)
        monkeypatch.delenv("SLS_POTRI_POOL", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = m.Context(0)
        h = m.Nll(c, X, 1)
        res[name] = h.gp_objective(y, x)
        assert c.prof_get("potrf_fallbacks")[1] == 0
        h.close(); c.close()
    assert res["fused_static"][0] == res["fused"][0] and np.array_equal(res["fused_static"][1], res["fused"][1])
    np.testing.assert_allclose(res["fused"][0], res["separate"][0], rtol=1e-11)
    g0 = res["separate"][1]
    np.testing.assert_allclose(res["fused"][1], g0, rtol=1e-7, atol=1e-9 * np.abs(g0).max())


def test_map_objective_handle_reused_with_changing_targets_and_query_staging(ctx, oracle, monkeypatch):
    """A MAP-objective handle uploads its targets only when they differ from what the device holds, reads the length scales from and
    writes its results into a mapped block it keeps: evaluations on ONE handle with y1, y2, y1 again, a value-only batch (which
    uploads y past the staging block) and y2 once more must equal the same evaluations on fresh handles, bit for bit; likewise a
    predict call on the page-locked staging path against SLS_IO_STAGE=0 (host transposition, pageable copies)."""
    m = sls()
    D, N = 6, 700
    X, y1, theta, b = synth_problem(oracle, D, N)
    y2 = y1[::-1].copy() * 1.5 + 0.1
    x1 = np.concatenate([[0.6, 0.01], np.linspace(0.4, 0.9, D)])
    x2 = x1 * 1.1
    def fresh(y, x):
        h = m.Nll(ctx, X, 1)
        r = h.gp_objective(y, x)
        h.close()
        return r
    want = {("y1", "x1"): fresh(y1, x1), ("y2", "x1"): fresh(y2, x1), ("y2", "x2"): fresh(y2, x2), ("y1", "x2"): fresh(y1, x2)}
    ys, xs = {"y1": y1, "y2": y2}, {"x1": x1, "x2": x2}
    h = m.Nll(ctx, X, 1)
    def check(yk, xk):
        v, g = h.gp_objective(ys[yk].copy(), xs[xk])       # a copy: the comparison is by content, not by address
        assert v == want[(yk, xk)][0], (yk, xk)
        np.testing.assert_array_equal(g, want[(yk, xk)][1])
    check("y1", "x1"); check("y2", "x1")                    # same factor (cached), new targets
    check("y2", "x2"); check("y1", "x2"); check("y1", "x1")
    batch = h.gp_objective_batch(y2, np.stack([x1, x2]))     # uploads y2 without the staging block
    np.testing.assert_allclose(batch, [want[("y2", "x1")][0], want[("y2", "x2")][0]], rtol=1e-9)
    check("y1", "x2"); check("y2", "x2")
    h.close()
    # query points through the staging block vs the pageable path
    M = 333
    Xs = synth_candidates(oracle, D, M)
    gp = m.GP(ctx, X, y1, theta, b, 1)
    mu1, sd1 = gp.predict(Xs)
    dm1, ds1 = gp.predict_grad(Xs)
    monkeypatch.setenv("SLS_IO_STAGE", "0")
    mu0, sd0 = gp.predict(Xs)
    dm0, ds0 = gp.predict_grad(Xs)
    monkeypatch.delenv("SLS_IO_STAGE")
    gp.close()
    for a_, b_ in ((mu1, mu0), (sd1, sd0), (dm1, dm0), (ds1, ds0)):
        np.testing.assert_array_equal(a_, b_)


def test_potrf_rejects_indefinite(ctx):
    A = np.eye(200)
    A[150, 150] = -1.0
    with pytest.raises(sls().SlsError):
        ctx.potrf(A)


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("D,N,M", [(1, 9, 33), (4, 60, 25), (8, 300, 200), (16, 512, 130)])
def test_gp_posterior_and_acquisition(ctx, oracle, kernel, D, N, M, path):
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, M)
    Xs[:, 0] = X[:, N // 2]                         # a candidate exactly on a data point (sigma^2 ~ b)
    ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    s = gp.summary()
    assert s["best_index"] == ref.predict_maximum_point_from_data()[0]
    close(gp.matrix(sls().GP_K_Y), oracle.calc_large_ky(kernel, X, theta, b), rtol=1e-10)
    Lo, _ = oracle.cholesky(oracle.calc_large_ky(kernel, X, theta, b))
    close(gp.matrix(sls().GP_CHOL_L), Lo, rtol=1e-7, atol=1e-10)
    close(s["logdet"], oracle.logdet_from_chol(Lo), rtol=1e-9)
    Kinv = gp.matrix(sls().GP_K_Y_INV)
    Kio = oracle.spd_inverse_from_chol(Lo)
    assert relerr(Kinv, Kio, floor=np.abs(Kio).max()) < 1e-8
    close(gp.matrix(sls().GP_ALPHA), oracle.chol_solve(Lo, y), rtol=1e-6, atol=1e-6 * np.abs(y).max())
    mu_o, sg_o = ref.predict_batch(Xs)
    mu, sg = gp.predict(Xs)
    close(mu, mu_o, rtol=RTOL, atol=1e-9)
    close(sg, sg_o, rtol=RTOL, atol=1e-9)
    dm_o, ds_o = ref.predict_grad_batch(Xs)
    dm, ds = gp.predict_grad(Xs)
    gscale = max(np.abs(dm_o).max(), 1e-12)
    close(dm, dm_o, rtol=RTOL, atol=1e-8 * gscale)
    close(ds, ds_o, rtol=RTOL, atol=1e-7 * np.abs(ds_o).max())
    for acq, h in ((0, 1.0), (1, 2.0)):
        v_o, g_o = ref.acq_eval_batch(Xs, acq, h)
        v, g = gp.acq_eval(Xs, acq, h)
        close(v, v_o, rtol=RTOL, atol=1e-9 * max(np.abs(v_o).max(), 1e-30))
        close(g, g_o, rtol=RTOL, atol=1e-7 * max(np.abs(g_o).max(), 1e-30))
        # value-only calls contract with L^-1 (var_gemm), value+gradient calls with K^-1: same number to rounding
        close(gp.acq_eval(Xs, acq, h, want_grad=False), v, rtol=1e-8, atol=1e-10 * max(np.abs(v).max(), 1e-30))
    gp.close()


def test_gp_against_mpmath_fixtures(ctx, fixtures, path):
    """The HIP path directly against the independent 50-digit answers (not via the oracle)."""
    for c in fixtures["gp_pipelines"]:
        X, Xs = np.array(c["X"]), np.array(c["Xs"])
        gp = sls().GP(ctx, X, c["y"], c["theta"], c["b"], c["kernel"])
        assert gp.summary()["best_index"] == c["best_index"]
        close(gp.summary()["mu_best"], c["mu_best"], rtol=1e-7)
        mu, sg = gp.predict(Xs)
        dm, ds = gp.predict_grad(Xs)
        ei, dei = gp.acq_eval(Xs, 0)
        ucb, ducb = gp.acq_eval(Xs, 1, 2.0)
        close(mu, c["mu"], rtol=2e-7, atol=1e-9)
        close(sg, c["sigma"], rtol=2e-7, atol=1e-9)
        close(dm.T, c["dmu"], rtol=2e-7, atol=1e-9)
        close(ds.T, c["dsigma"], rtol=2e-6, atol=1e-8)
        close(ei, c["ei"], rtol=2e-6, atol=1e-10)
        close(dei.T, c["dei"], rtol=2e-6, atol=1e-9)
        close(ucb, c["ucb"], rtol=2e-7, atol=1e-9)
        close(ducb.T, c["ducb"], rtol=2e-6, atol=1e-8)
        gp.close()


@pytest.mark.parametrize("name", [f"k{k}_N{n}" for n in (130, 300, 512, 2048) for k in (0, 1)])
def test_gp_against_scipy_fixtures(ctx, scipy_cases, name, path):
    """The HIP path directly against the scipy/LAPACK pipelines at N = 130 .. 2048 (tests/golden/scipy_pipelines.npz):
    factor, inverse, alpha, log-determinant, arg max, and every predictive quantity.  N <= 512 runs on both the per-point
    wavefront kernels and the tiled MFMA pipeline; N = 2048 is tiled only."""
    c = scipy_cases[name]
    X, y, theta, b, Xs, kernel = c["X"], c["y"], c["theta"], float(c["b"]), c["Xs"], int(c["kernel"])
    if X.shape[1] > 512 and path == "wave":
        pytest.skip("the wavefront kernels serve N <= 512")
    m = sls()
    gp = m.GP(ctx, X, y, theta, b, kernel)
    s = gp.summary()
    assert s["best_index"] == int(c["best_index"])
    close(s["mu_best"], c["mu_best"], rtol=1e-7)
    close(s["logdet"], c["logdet"], rtol=1e-10)
    rows = c["rows"]
    L = gp.matrix(m.GP_CHOL_L)
    close(np.diag(L), c["L_diag"], rtol=1e-9)
    close(L[rows], c["L_rows"], rtol=1e-7, atol=1e-11)
    Kinv = gp.matrix(m.GP_K_Y_INV)
    scale = np.abs(c["Kinv_diag"]).max()
    close(np.diag(Kinv), c["Kinv_diag"], rtol=1e-7)
    close(Kinv[rows], c["Kinv_rows"], rtol=1e-6, atol=1e-8 * scale)
    close(gp.matrix(m.GP_ALPHA), c["alpha"], rtol=1e-6, atol=1e-8 * np.abs(c["alpha"]).max())
    mu, sg = gp.predict(Xs)
    dmu, dsg = gp.predict_grad(Xs)
    ei, dei = gp.acq_eval(Xs, 0)
    ucb, ducb = gp.acq_eval(Xs, 1, scipy_cases["_ucb_h"])
    close(mu, c["mu"], rtol=RTOL, atol=1e-9)
    close(sg, c["sigma"], rtol=RTOL, atol=1e-9)
    close(dmu, c["dmu"], rtol=RTOL, atol=1e-8 * np.abs(c["dmu"]).max())
    close(dsg, c["dsigma"], rtol=RTOL, atol=1e-7 * np.abs(c["dsigma"]).max())
    close(ei, c["ei"], rtol=RTOL, atol=1e-8 * np.abs(c["ei"]).max())
    close(dei, c["dei"], rtol=RTOL, atol=1e-7 * np.abs(c["dei"]).max())
    close(ucb, c["ucb"], rtol=RTOL, atol=1e-9)
    close(ducb, c["ducb"], rtol=RTOL, atol=1e-7 * np.abs(c["ducb"]).max())
    gp.close()


def test_chunked_evaluation_matches_single_pass(ctx, oracle, monkeypatch):
    monkeypatch.setenv("SLS_WAVE_PATH", "0")          # the chunk loop belongs to the tiled pipeline
    X, y, theta, b = synth_problem(oracle, 6, 200)
    Xs = synth_candidates(oracle, 6, 700)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    v1, g1 = gp.acq_eval(Xs)
    ctx.set_candidate_chunk(256)
    v2, g2 = gp.acq_eval(Xs)
    ctx.set_candidate_chunk(16384)
    assert np.array_equal(v1, v2) and np.array_equal(g1, g2)
    gp.close()


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("acq", [0, 1])
def test_multistart_maximizer_matches_oracle(ctx, oracle, kernel, acq, path):
    D, N, S, n_local = 5, 120, 96, 25
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    ro = ref.acq_maximize(starts, n_local, acq, 2.0, diag=True)
    rg = gp.acq_maximize(starts, n_local, acq, 2.0)
    # every start follows the same trajectory (same algorithm, fp64 rounding apart); the rare exception is a start whose
    # Armijo test sat at rounding level of its threshold
    assert_starts_agree(rg, ro, label=f"acq_maximize D={D} N={N} S={S} kernel={kernel} acq={acq}", max_divergent=3)   # measured: 0 or 1 of 96
    # many starts converge to the same maximiser, so the winning INDEX is decided by the last bits; what must
    # agree is the chosen maximiser and its value (north_star: within 1e-6 relative), and the GPU's winner must
    # be one of the oracle's tied winners
    assert ro["y_stars"][rg["index"]] >= ro["value"] * (1 - 1e-9) - 1e-300
    close(rg["value"], ro["value"], rtol=RTOL)
    close(rg["x"], ro["x"], rtol=RTOL, atol=1e-7)
    assert np.all((rg["x_stars"] >= 0) & (rg["x_stars"] <= 1))
    v0 = gp.acq_eval(starts, acq, 2.0, want_grad=False)
    assert np.all(rg["y_stars"] >= v0 - 1e-9 * np.abs(v0).max())
    gp.close()


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("D,N,M", [(8, 640, 1000), (64, 1024, 5000), (33, 384, 130)])
def test_gradient_contraction_split_is_bit_identical(ctx, oracle, kernel, D, N, M, monkeypatch):
    """grad_gemm64_kernel sums the contraction over the training points as four quarter ranges, ((q0 + q1) + q2) + q3: by one
    workgroup per tile (SLS_GRAD_SPLIT_TILES=0) or by four workgroups + grad_reduce4_kernel (default).  Same bits."""
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, M)
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    out = []
    for flag in (None, "0"):
        if flag is None:
            monkeypatch.delenv("SLS_GRAD_SPLIT_TILES", raising=False)
        else:
            monkeypatch.setenv("SLS_GRAD_SPLIT_TILES", flag)
        out.append(gp.acq_eval(Xs, 0, 1.0, want_grad=True))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    gp.close()


@pytest.mark.parametrize("D,N,S,n_local,history", [(6, 300, 700, 40, 6), (16, 640, 3000, 30, 3), (37, 384, 2000, 30, 8), (64, 512, 1500, 25, 6),
                                                    (70, 256, 600, 12, 6)])
def test_lbfgs_step_register_form_is_bit_identical(ctx, oracle, D, N, S, n_local, history, monkeypatch):
    """lbfgs_step_reg_kernel (a start's vectors in registers, D <= 64) against lbfgs_step_kernel (every pass through global
    memory, SLS_LBFGS_REG=0; also what D > 64 runs): same expressions in the same order, so every start must end with the same
    bits -- accepted and rejected steps, curvature pairs that are dropped, resets, backtracking, corner starts."""
    monkeypatch.setenv("SLS_WAVE_PATH", "0")
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    starts[:, ::5] = np.round(starts[:, ::5])
    gp = sls().GP(ctx, X, y, theta, b, 1)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SLS_LBFGS_REG", flag)
        r = gp.acq_maximize(starts, n_local, opts=sls().LbfgsOpts(history, 1e-4, 0.5, 0.0, 20))
        out[flag] = (r["y_stars"], r["x_stars"], r["value"], r["index"], r["x"], gp.last_stats()["evals_issued"])
    for va, vu in zip(out["1"], out["0"]):
        assert np.array_equal(np.asarray(va), np.asarray(vu))
    gp.close()


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("D,N,S,n_local,pair", [(6, 300, 700, 30, False), (16, 2048, 9000, 14, False), (4, 200, 300, 20, True)])
def test_active_set_compaction_is_bit_identical(ctx, oracle, kernel, D, N, S, n_local, pair, monkeypatch):
    """The lock-step maximiser drops finished starts and compacts the live ones into dense tiles every round (default).
    Every start must end with exactly the bits of the uncompacted schedule (SLS_COMPACT=0: all S starts re-evaluated every
    round), for the single-regressor objective and for the pair objective of FindNextPoints; N = 2048 with 9000 starts
    crosses from the persistent generation-gated acq_gemm (>= 1024 tiles) to its one-tile-per-workgroup form as the
    active set shrinks."""
    monkeypatch.setenv("SLS_WAVE_PATH", "0")
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    starts[:, ::7] = np.round(starts[:, ::7])            # corner starts: many stop after a few evaluations
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    g2 = None
    if pair:
        extra = synth_candidates(oracle, D, 3, seed=77)
        g2 = sls().GP(ctx, np.concatenate([X, extra], axis=1), np.concatenate([y, [0.2, 0.1, 0.3]]), theta, b, kernel)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SLS_COMPACT", flag)
        if pair:
            r = gp.acq_maximize_pair(g2, starts, n_local)
            out[flag] = (r["value"], r["index"], r["x"], gp.last_stats())
        else:
            r = gp.acq_maximize(starts, n_local)
            out[flag] = (r["y_stars"], r["x_stars"], r["value"], r["index"], r["x"], gp.last_stats())
    a, u = out["1"], out["0"]
    for va, vu in zip(a[:-1], u[:-1]):
        assert np.array_equal(np.asarray(va), np.asarray(vu))
    sa, su = a[-1], u[-1]
    # evals_issued counts evaluations of starts that were still moving -- also where SLS_COMPACT=0 re-evaluates the finished
    # ones: the same starts retire in the same round under both schedules
    assert su["evals_cap"] == sa["evals_cap"] == S * n_local
    assert S <= sa["evals_issued"] == su["evals_issued"] <= S * n_local
    assert sa["rounds"] <= n_local
    if not pair:
        assert sa["evals_issued"] < S * n_local             # the corner starts retire early
        # same end points as the oracle's all-starts-every-round loop
        ro = oracle.Regressor(X, y, theta, b, kernel=kernel).acq_maximize(starts, n_local, diag=True) if N <= 300 else None
        if ro is not None:
            # The one start on record here (N = 300, Matern, start 50: 8.2e-7 of the largest end value, 1.1e-6 of its own, from the
            # oracle; Armijo margin 1.1e-2) was examined on the CPU in round 5 (profiles/r05_start50_probe.log): no discrete
            # decision flips -- the trajectory bounces between faces of the box (its active set changes in 14 of the 30 rounds, it
            # sits in a corner with all six bounds active at round 3) and is still climbing steeply when the budget ends (0.0796 ->
            # 0.0814 in the last round); the ORACLE's own end value of this start moves by 2.4e-7 under one ulp of the start and by
            # 7.0e-7 under 256 ulps of the signal variance, not monotonically.  The former "same basin" escape is replaced by that
            # measurement: the gap must be within twice the band the oracle itself shows.
            def probe(i):
                return oracle_end_value_sensitivity(oracle, X, y, theta, b, kernel, starts, i, n_local, 0, 1.0, ro)
            assert_starts_agree(dict(y_stars=a[0]), ro, label=f"compaction N={N} kernel={kernel}", ulp_probe=probe, max_divergent=2)
            close(a[2], ro["value"], rtol=RTOL)
    gp.close()
    if g2 is not None:
        g2.close()


def test_maximizer_1d_demo_scenario(ctx, oracle):
    """demos/bayesian_optimization_1d/core.cpp:70-73: f(x) = 1 - 1.5 x sin(13 x); EI maximiser found on [0,1]."""
    rng = np.random.default_rng(1)
    X = rng.uniform(0, 1, (1, 8))
    y = 1.0 - 1.5 * X[0] * np.sin(13.0 * X[0])
    gp = sls().GP(ctx, X, y, [0.5, 0.15], 1e-4, 1)
    ref = oracle.Regressor(X, y, [0.5, 0.15], 1e-4, kernel=1)
    starts = rng.uniform(0, 1, (1, 64))
    rg, ro = gp.acq_maximize(starts, 30), ref.acq_maximize(starts, 30)
    assert ro["y_stars"][rg["index"]] >= ro["value"] * (1 - 1e-9)
    close(rg["value"], ro["value"], rtol=RTOL)
    close(rg["x"], ro["x"], rtol=RTOL, atol=1e-7)
    grid = np.linspace(0, 1, 2001)[None, :]
    assert rg["value"] >= gp.acq_eval(grid, want_grad=False).max() * (1 - 1e-3)
    gp.close()


def test_start_offset_and_rank_merge(ctx, oracle):
    """Sharding the start set over ranks and merging (value, global index) reproduces the single-rank answer."""
    D, N, S = 3, 40, 64
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    full = gp.acq_maximize(starts, 15)
    parts = []
    for r in range(4):
        lo, hi = r * S // 4, (r + 1) * S // 4
        p = gp.acq_maximize(starts[:, lo:hi], 15, offset=lo, want_all=False)
        parts.append((p["value"], p["index"], p["x"]))
    v, i, x = sls().merge_rank_results(parts)
    assert i == full["index"] and v == full["value"] and np.array_equal(x, full["x"])
    gp.close()


def test_invalid_arguments(ctx):
    m = sls()
    with pytest.raises(m.SlsError):
        m.GP(ctx, np.zeros((2, 3)), np.zeros(3), [0.5, -1.0, 0.5], 0.01)
    with pytest.raises(m.SlsError):
        m.GP(ctx, np.zeros((2, 3)), np.zeros(3), [0.5, 0.5, 0.5], 0.01, kernel=7)


# ---- MAP objectives ------------------------------------------------------------

@pytest.fixture(params=["small", "tiled"])
def nllpath(request, monkeypatch):
    """N <= 128 evaluations run the fused single-launch kernel (kernels_small.hip) unless SLS_NLL_SMALL=0: both paths must pass."""
    monkeypatch.setenv("SLS_NLL_SMALL", "1" if request.param == "small" else "0")
    return request.param


def test_gp_map_objective_against_mpmath_and_oracle(ctx, oracle, fixtures, nllpath):
    for c in fixtures["gp_map"]:                      # independent 50-digit values
        h = sls().Nll(ctx, np.array(c["X"]), c["kernel"])
        v, g = h.gp_objective(c["y"], c["x"])
        close(v, c["value"], rtol=1e-9)
        close(g, c["grad"], rtol=1e-6, atol=1e-7)
        close(h.gp_objective(c["y"], c["x"], want_grad=False), v, rtol=0)
        h.close()
    for kernel in (0, 1):                             # the reference's tensor + trace formulation, larger N
        D, N = 7, 200
        X, y, theta, b = synth_problem(oracle, D, N)
        x = np.concatenate([[0.6, 0.01], np.linspace(0.35, 0.8, D)])
        vo, go = oracle.gp_map_objective(kernel, X, y, x, as_written=True)
        h = sls().Nll(ctx, X, kernel)
        v, g = h.gp_objective(y, x)
        close(v, vo, rtol=1e-9)
        close(g, go, rtol=RTOL, atol=1e-6 * np.abs(go).max())
        h.close()


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("use_map", [False, True])
def test_preference_objective(ctx, oracle, fixtures, kernel, use_map, nllpath):
    for c in fixtures["pref_objective"]:
        if c["kernel"] != kernel or c["use_map"] != use_map:
            continue
        h = sls().Nll(ctx, np.array(c["X"]), kernel)
        v, g = h.pref_objective(c["prefs"], c["x"], use_map=use_map)
        close(v, c["value"], rtol=1e-9)
        close(g, c["grad"], rtol=1e-6, atol=1e-6)
        h.close()
    # sequential-line-search style data: M points, one preference tuple per "slider" observation
    rng = np.random.default_rng(7)
    D, M = 6, 45
    X = rng.uniform(0, 1, (D, M))
    prefs = [[3 * i, 3 * i + 1, 3 * i + 2] for i in range(M // 3)]
    yv = rng.normal(0, 0.02, M)
    x = np.concatenate([yv, [0.45, 0.004], rng.uniform(0.4, 0.6, D)]) if use_map else yv
    vo, go = oracle.pref_objective(kernel, X, prefs, x, use_map=use_map)
    h = sls().Nll(ctx, X, kernel)
    v, g = h.pref_objective(prefs, x, use_map=use_map)
    close(v, vo, rtol=1e-9)
    close(g, go, rtol=RTOL, atol=1e-6 * np.abs(go).max())
    # cached factorisation path (same theta, new y) must agree with a fresh handle
    x2 = x.copy(); x2[:M] += 0.001
    v2, g2 = h.pref_objective(prefs, x2, use_map=use_map)
    vo2, go2 = oracle.pref_objective(kernel, X, prefs, x2, use_map=use_map)
    close(v2, vo2, rtol=1e-9)
    close(g2, go2, rtol=RTOL, atol=1e-6 * np.abs(go2).max())
    h.close()


def test_nll_core_terms(ctx, oracle):
    D, N = 5, 150
    X, y, theta, b = synth_problem(oracle, D, N)
    K = oracle.calc_large_ky(1, X, theta, b)
    L, _ = oracle.cholesky(K)
    h = sls().Nll(ctx, X, 1)
    r = h.eval(y, theta, b)
    al = oracle.chol_solve(L, y)
    close(r["alpha"], al, rtol=1e-7, atol=1e-9)
    close(r["quad"], y @ al, rtol=1e-9)
    close(r["logdet"], oracle.logdet_from_chol(L), rtol=1e-10)
    Kinv = oracle.spd_inverse_from_chol(L)
    close(r["grad_b"], 0.5 * (al @ al - np.trace(Kinv)), rtol=1e-7)
    h.close()


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("D,N", [(1, 1), (1, 5), (2, 16), (3, 17), (16, 64), (5, 100), (4, 128), (32, 90)])
def test_fused_small_objective_matches_tiled_and_oracle(ctx, oracle, kernel, D, N, monkeypatch):
    """kernels_small.hip (whole evaluation in one workgroup) against the tiled pipeline and the oracle, over the 16-block
    boundaries of its LDS factorisation (N = 16, 17, 128) and with D above its gradient limit (D = 32: no theta gradient)."""
    X, y, theta, b = synth_problem(oracle, D, N)
    K = oracle.calc_large_ky(kernel, X, theta, b)
    L, _ = oracle.cholesky(K)
    al = oracle.chol_solve(L, y)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SLS_NLL_SMALL", flag)
        h = sls().Nll(ctx, X, kernel)
        r = h.eval(y, theta, b, want_grad=(D <= 16))
        close(r["alpha"], al, rtol=1e-7, atol=1e-9 * max(np.abs(al).max(), 1e-30))
        close(r["quad"], y @ al, rtol=1e-9)
        close(r["logdet"], oracle.logdet_from_chol(L), rtol=1e-10, atol=1e-12)
        if D <= 16:
            x = np.concatenate([[theta[0], b], theta[1:]])
            vo, go = oracle.gp_map_objective(kernel, X, y, x)
            v, g = h.gp_objective(y, x)
            close(v, vo, rtol=1e-9, atol=1e-10)
            close(g, go, rtol=RTOL, atol=1e-6 * max(np.abs(go).max(), 1e-30))
            res[flag] = (r["quad"], r["logdet"], r["grad_b"], r["grad_theta"], v, g)
        else:
            res[flag] = (r["quad"], r["logdet"])
        h.close()
    for u, w in zip(res["1"], res["0"]):
        close(u, w, rtol=1e-7, atol=1e-9 * max(np.abs(np.asarray(w)).max(), 1e-30))


# ---- edge cases (empty / ragged / maximum sizes / duplicates) ----------------------------------------------

def test_single_data_point_and_single_candidate(ctx, oracle, path):
    X = np.array([[0.3], [0.7]])
    y = np.array([1.25])
    theta = np.array([0.5, 0.4, 0.6])
    for kernel in (0, 1):
        gp = sls().GP(ctx, X, y, theta, 0.01, kernel)
        ref = oracle.Regressor(X, y, theta, 0.01, kernel=kernel)
        xs = np.array([[0.31], [0.65]])
        close(gp.predict(xs)[0], ref.predict_batch(xs)[0], rtol=1e-10)
        close(gp.predict(xs)[1], ref.predict_batch(xs)[1], rtol=1e-8)
        v, g = gp.acq_eval(xs, 1, 1.5)
        vo, go = ref.acq_eval_batch(xs, 1, 1.5)
        close(v, vo, rtol=1e-9)
        close(g, go, rtol=1e-7, atol=1e-12)
        assert gp.summary()["best_index"] == 0
        r = gp.acq_maximize(xs, 5, 1, 1.5)
        assert r["index"] == 0 and r["x"].shape == (2,)
        gp.close()


def test_duplicate_and_near_duplicate_training_points(ctx, oracle):
    """The reference merges points closer than 1e-4 only in the data manager; the regressors must still cope with exact
    duplicates (K_y stays SPD through the noise term)."""
    D, N = 3, 40
    X, y, theta, b = synth_problem(oracle, D, N)
    X[:, 7] = X[:, 3]
    X[:, 9] = X[:, 3] + 1e-9
    y[7] = y[3]
    for kernel in (0, 1):
        gp = sls().GP(ctx, X, y, theta, b, kernel)
        ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
        Xs = synth_candidates(oracle, D, 50)
        Xs[:, 0] = X[:, 3]
        close(gp.predict(Xs)[0], ref.predict_batch(Xs)[0], rtol=1e-6, atol=1e-8)
        close(gp.predict(Xs)[1], ref.predict_batch(Xs)[1], rtol=1e-5, atol=1e-8)
        gp.close()


def test_noiseless_formulation_b_zero(ctx, oracle):
    """SEQUENTIAL_LINE_SEARCH_USE_NOISELESS_FORMULATION: b = 0, well-separated points -> interpolation, sigma(x_i) ~ 0."""
    D, N = 2, 25
    g = np.linspace(0.05, 0.95, 5)
    X = np.array([[a, c] for a in g for c in g]).T.copy()
    y = np.sin(3 * X[0]) * np.cos(2 * X[1])
    theta = np.array([0.5, 0.15, 0.15])
    gp = sls().GP(ctx, X, y, theta, 0.0, 1)
    mu, sg = gp.predict(X)
    close(mu, y, rtol=1e-7, atol=1e-8)
    assert np.all(sg < 1e-4)
    ei, dei = gp.acq_eval(X)               # EI and its gradient are finite (zero) where sigma < 1e-10 or tiny
    assert np.all(np.isfinite(ei)) and np.all(ei >= 0)
    ref = oracle.Regressor(X, y, theta, 0.0, kernel=1)
    Xs = synth_candidates(oracle, D, 40)
    close(gp.predict(Xs)[0], ref.predict_batch(Xs)[0], rtol=1e-6, atol=1e-8)
    gp.close()


def test_candidates_outside_the_unit_box_and_ragged_counts(ctx, oracle, path):
    D, N = 4, 33
    X, y, theta, b = synth_problem(oracle, D, N)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    ref = oracle.Regressor(X, y, theta, b, kernel=1)
    for M in (1, 127, 128, 129, 257):
        Xs = synth_candidates(oracle, D, M, seed=M) * 1.6 - 0.3      # predictions are defined on all of R^D
        close(gp.predict(Xs)[0], ref.predict_batch(Xs)[0], rtol=1e-6, atol=1e-9)
        v, g = gp.acq_eval(Xs)
        vo, go = ref.acq_eval_batch(Xs)
        close(v, vo, rtol=1e-6, atol=1e-12)
        close(g, go, rtol=1e-6, atol=1e-9 * max(np.abs(go).max(), 1e-30))
    # the maximiser clamps its starts to the box first (reference bounds [0,1]^D, acquisition-function.cpp:118-119)
    starts = synth_candidates(oracle, D, 30) * 1.6 - 0.3
    r, ro = gp.acq_maximize(starts, 8), ref.acq_maximize(starts, 8)
    assert np.all((r["x_stars"] >= 0) & (r["x_stars"] <= 1))
    close(r["value"], ro["value"], rtol=1e-6)
    gp.close()


def test_high_dimension_and_ard(ctx, oracle, path):
    D, N, M = 128, 150, 64
    X, y, theta, b = synth_problem(oracle, D, N)
    theta[1:] *= np.linspace(0.5, 2.0, D)
    for kernel in (0, 1):
        gp = sls().GP(ctx, X, y, theta, b, kernel)
        ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
        Xs = synth_candidates(oracle, D, M)
        v, g = gp.acq_eval(Xs, 1, 2.0)
        vo, go = ref.acq_eval_batch(Xs, 1, 2.0)
        close(v, vo, rtol=1e-6)
        close(g, go, rtol=1e-6, atol=1e-8 * np.abs(go).max())
        gp.close()


def test_expected_improvement_flat_region_is_exactly_zero(ctx, oracle):
    """Far from the data with a short length scale EI underflows to 0 with a zero gradient; such starts stop at once
    and the maximiser returns them unchanged (value 0), like the oracle."""
    X = np.array([[0.1, 0.12, 0.15]])
    y = np.array([1.0, 3.0, 2.0])
    theta = np.array([1e-3, 0.01])
    gp = sls().GP(ctx, X, y, theta, 1e-6, 0)
    ref = oracle.Regressor(X, y, theta, 1e-6, kernel=0)
    xs = np.array([[0.9, 0.95, 0.5]])
    v, g = gp.acq_eval(xs)
    vo, go = ref.acq_eval_batch(xs)
    assert np.array_equal(v, vo) and np.all(v == 0.0) and np.all(g == 0.0) and np.all(go == 0.0)
    r, ro = gp.acq_maximize(xs, 10), ref.acq_maximize(xs, 10)
    assert np.array_equal(r["x_stars"], xs) and r["value"] == 0.0 and ro["value"] == 0.0 and r["index"] == ro["index"] == 0
    gp.close()


def test_lbfgs_options_and_single_evaluation(ctx, oracle):
    D, N, S = 3, 30, 40
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    v0 = gp.acq_eval(starts, want_grad=False)
    r1 = gp.acq_maximize(starts, 1)                      # n_local = 1: the starts themselves
    # (small problems run the per-start wavefront kernel, the batched evaluation the tiled MFMA kernels: same numbers
    #  up to summation order)
    close(r1["y_stars"], v0, rtol=1e-12, atol=1e-300)
    assert r1["index"] == int(np.argmax(r1["y_stars"]))
    opts = sls().LbfgsOpts(3, 1e-4, 0.5, 0.0, 20)        # shorter memory: still monotone and inside the box
    r = gp.acq_maximize(starts, 12, opts=opts)
    assert np.all(r["y_stars"] >= v0 - 1e-13 * np.abs(v0).max())
    with pytest.raises(sls().SlsError):
        gp.acq_maximize(starts, 5, opts=sls().LbfgsOpts(9, 1e-4, 0.5, 0.0, 20))
    gp.close()


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("N0", [5, 126, 128, 200])
def test_append_point_equals_refit(ctx, oracle, kernel, N0):
    """sls_gp_append_point (rank-1 growth, used by FindNextPoints) against a fresh fit on the extended data; N0 = 126 / 128
    cross the 128-padding boundary (in-place update vs buffer growth)."""
    D, extra = 3, 4
    X, y, theta, b = synth_problem(oracle, D, N0 + extra)
    gp = sls().GP(ctx, X[:, :N0], y[:N0], theta, b, kernel)
    for i in range(N0, N0 + extra):
        gp.append_point(X[:, i], y[i])
    ref = sls().GP(ctx, X, y, theta, b, kernel)
    m = sls()
    Ki, Kr = gp.matrix(m.GP_K_Y_INV), ref.matrix(m.GP_K_Y_INV)
    assert relerr(Ki, Kr, floor=np.abs(Kr).max()) < 1e-9
    close(gp.matrix(m.GP_CHOL_L), ref.matrix(m.GP_CHOL_L), rtol=1e-8, atol=1e-10)
    close(gp.matrix(m.GP_ALPHA), ref.matrix(m.GP_ALPHA), rtol=1e-6, atol=1e-7 * np.abs(y).max())
    sa, sb = gp.summary(), ref.summary()
    assert sa["best_index"] == sb["best_index"]
    close(sa["logdet"], sb["logdet"], rtol=1e-10)
    close(sa["mu_best"], sb["mu_best"], rtol=1e-8)
    Xs = synth_candidates(oracle, D, 33)
    for a_, b_ in zip(gp.acq_eval(Xs), ref.acq_eval(Xs)):
        close(a_, b_, rtol=1e-6, atol=1e-9 * max(np.abs(b_).max(), 1e-30))
    oref = oracle.Regressor(X, y, theta, b, kernel=kernel)
    close(gp.predict(Xs)[1], oref.predict_batch(Xs)[1], rtol=1e-6, atol=1e-9)
    gp.close(); ref.close()


@pytest.mark.parametrize("acq", [0, 1])
def test_pair_objective_of_find_next_points(ctx, oracle, acq):
    """objective_for_multiple_points (src/acquisition-function.cpp:63-110): mu, mu+ from the original regressor, sigma from
    the variance-updated dummy regressor.  Composed here from two oracle regressors with the mathtoolbox EI formula."""
    from scipy.special import erfc
    D, N = 3, 35
    X, y, theta, b = synth_problem(oracle, D, N)
    extra = synth_candidates(oracle, D, 2, seed=99)
    X2 = np.concatenate([X, extra], axis=1)
    y2 = np.concatenate([y, [0.3, 0.1]])
    g1 = sls().GP(ctx, X, y, theta, b, 1)
    g2 = sls().GP(ctx, X2, y2, theta, b, 1)
    r1 = oracle.Regressor(X, y, theta, b, kernel=1)
    r2 = oracle.Regressor(X2, y2, theta, b, kernel=1)
    Xs = synth_candidates(oracle, D, 70)
    mu, _ = r1.predict_batch(Xs)
    _, sg = r2.predict_batch(Xs)
    dmu, _ = r1.predict_grad_batch(Xs)
    _, dsg = r2.predict_grad_batch(Xs)
    if acq == 0:
        mu_best = r1.predict_batch(r1.predict_maximum_point_from_data()[1][:, None])[0][0]
        u = (mu - mu_best) / sg
        Phi, phi = 0.5 * erfc(-u / np.sqrt(2)), np.exp(-0.5 * u * u) / np.sqrt(2 * np.pi)
        vo, go = (mu - mu_best) * Phi + sg * phi, Phi * dmu + phi * dsg
    else:
        vo, go = mu + 1.7 * sg, dmu + 1.7 * dsg
    v, g = g1.acq_eval_pair(g2, Xs, acq, 1.7)
    close(v, vo, rtol=RTOL, atol=1e-12)
    close(g, go, rtol=RTOL, atol=1e-9 * np.abs(go).max())
    # near the appended points the dummy regressor's deviation collapses, so the pair objective avoids them
    ve = g1.acq_eval_pair(g2, extra, acq, 1.7, want_grad=False)
    vs = g1.acq_eval(extra, acq, 1.7, want_grad=False)
    assert np.all(ve < vs)
    r = g1.acq_maximize_pair(g2, Xs[:, :32], 10, acq, 1.7)
    assert r["value"] >= v[:32].max() - 1e-15
    g1.close(); g2.close()


def _within(x, ref, rtol, atol):
    x, ref = np.asarray(x, dtype=float), np.asarray(ref, dtype=float)
    return bool(np.all(np.abs(x - ref) <= rtol * np.abs(ref) + atol))


@pytest.mark.parametrize("seed", range(10 + int(os.environ.get("SLS_TEST_EXTRA_SEEDS", "0"))))   # extra seeds: one-off stress sweeps
def test_randomised_configurations(ctx, oracle, seed, path):
    """Random problem shapes / hyper-parameters (ragged sizes, anisotropic length scales, noise from 1e-6 to 1e-1).

    north_star's 1e-6 holds FLAT for the 20 committed cases (seeds 0-9 on both paths: measured max 2.6e-11 on sigma, 1.4e-12 on
    EI / UCB, 2.2e-12 on gradients, kappa up to 7e7).  It cannot hold for every input: sigma^2 = a - k^T K_y^-1 k is a
    cancellation, and with the reference's formula (explicit inverse, src/gaussian-process-regressor.cpp:245-255) its fp64
    error is of the order cond(K_y) eps a, i.e. cond eps a / (2 sigma) on sigma.  A sweep of 490 more seeds
    (SLS_TEST_EXTRA_SEEDS=490, profiles/r03_random_sweep.json) has such cases, e.g. seed 46: sigma down to 1e-3 at cond 2.6e6,
    seed 67: sigma down to 9e-4 at cond 6.5e7 -- where the oracle (the reference's formula) AND the HIP path (explicit L^-1 /
    K_y^-1 as well) are each off by 1e-8 / 1e-6 against an extended-precision solve, LAPACK's solve-based evaluation by 1e-12
    (tools/seed_conditioning.py, profiles/r03_seed_conditioning.log); in 500 seeds, 12 are of this kind, all with D <= 5 and
    sigma_min <= 3e-3.  Every case must therefore agree within 1e-6 PLUS that first-order bound (which is zero to working
    precision unless the problem is ill-conditioned AND sigma is tiny); the record says which cases needed the second term."""
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.integers(1, 40)); N = int(rng.integers(2, 400)); M = int(rng.integers(1, 300)); kernel = int(rng.integers(0, 2))
    X = rng.uniform(0, 1, (D, N))
    y = np.sin(X.sum(axis=0) * rng.uniform(1, 4)) + 0.05 * rng.normal(size=N)
    theta = np.concatenate([[rng.uniform(0.1, 2.0)], rng.uniform(0.2, 1.5, D) * np.sqrt(max(D, 4) / 4.0)])
    b = float(10 ** rng.uniform(-6, -1))
    Xs = rng.uniform(-0.1, 1.1, (D, M))
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
    mu, sg = gp.predict(Xs); muo, sgo = ref.predict_batch(Xs)
    scale = max(np.abs(muo).max(), 1e-30)
    eps = np.finfo(float).eps
    kappa = (theta[0] * N + b) / b                         # >= cond(K_y): eigenvalues in [b, a N + b]
    d_s2 = kappa * eps * theta[0]                          # first-order fp64 error of k^T K_y^-1 k through an explicit inverse
    d_sigma = d_s2 / (2.0 * np.maximum(sgo, 1e-150))       # ... of sigma, per point
    d_mu = kappa * eps * np.abs(y).max()
    strict = True                                          # every comparison within the flat 1e-6?
    strict &= _within(mu, muo, RTOL, 1e-6 * scale)
    assert _within(mu, muo, RTOL, 1e-6 * scale + d_mu)
    a_sig = 1e-7 * np.sqrt(theta[0])
    strict &= _within(sg, sgo, 1e-6, a_sig)
    assert _within(sg, sgo, 1e-6, a_sig + d_sigma), (np.max(np.abs(sg - sgo)), kappa)
    errs = dict(sigma=relerr(sg, np.maximum(sgo, 1e-7 * np.sqrt(theta[0]))))
    for acq, h in ((0, 1.0), (1, 0.7)):
        v, g = gp.acq_eval(Xs, acq, h); vo, go = ref.acq_eval_batch(Xs, acq, h)
        a_v = 1e-7 * max(np.abs(vo).max(), 1e-30)
        d_v = (0.4 * d_sigma + d_mu) if acq == 0 else (d_mu + h * d_sigma)      # |dEI/dsigma| = phi <= 0.4, |dEI/dmu| = Phi <= 1
        strict &= _within(v, vo, 1e-6, a_v)
        assert _within(v, vo, 1e-6, a_v + d_v), (acq, np.max(np.abs(v - vo)), kappa)
        finite = np.isfinite(go)
        assert np.array_equal(np.isfinite(g), finite)
        gmax = max(np.abs(go[finite]).max(), 1e-30) if finite.any() else 1.0
        # gradients carry 1 / sigma and, through phi(z) and Phi(z), z = (mu - mu+) / sigma: the relative bound of sigma per column
        # times (1 + z^2) <~ 50, on the column's largest component
        colmax = np.max(np.abs(np.where(finite, go, 0.0)), axis=0, keepdims=True)
        d_g = 50.0 * (d_sigma / np.maximum(sgo, 1e-150))[None, :] * colmax + np.zeros_like(go)
        strict &= _within(g[finite], go[finite], 1e-6, 1e-6 * gmax)
        assert _within(g[finite], go[finite], 1e-6, 1e-6 * gmax + d_g[finite]), (acq, kappa)
        errs[f"acq{acq}"] = float(np.max(np.abs(v - vo)) / max(np.abs(vo).max(), 1e-30))
        errs[f"grad{acq}"] = float(np.max(np.abs(g[finite] - go[finite])) / gmax) if finite.any() else 0.0
    if seed < 10:
        assert strict, "a committed case no longer holds at the flat 1e-6"
    from util import record
    record("randomised", seed=int(seed), path=path, D=D, N=N, M=M, kernel=kernel, b=b, kappa=float(kappa), flat_1e6=bool(strict),
           sigma_min=float(np.min(sgo)), bound_sigma_rel=float(np.max(d_sigma / np.maximum(sgo, 1e-150))), **errs)
    gp.close()


@pytest.mark.parametrize("D,N,S,n_local", [(1, 20, 1, 40), (32, 61, 1, 320), (32, 128, 7, 30), (8, 100, 256, 12), (70, 130, 3, 25), (6, 300, 40, 10)])
def test_wave_path_lds_staging_is_bit_identical(ctx, oracle, D, N, S, n_local, monkeypatch):
    """Launches of at most 256 starts copy K^-1 (and, if it still fits, the design matrix) into LDS once per workgroup
    (kernels_wave.hip): values only move, so every start must end with the bits of the unstaged kernel (SLS_WAVE_STAGE=0) --
    with both arrays staged, with K^-1 alone (N = 128: the design matrix no longer fits) and with neither (N = 300)."""
    monkeypatch.setenv("SLS_WAVE_PATH", "1")
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SLS_WAVE_STAGE", flag)
        r = gp.acq_maximize(starts, n_local)
        out[flag] = (r["y_stars"], r["x_stars"], r["value"], r["index"], r["x"], gp.last_stats()["evals_issued"])
    for va, vu in zip(out["1"], out["0"]):
        assert np.array_equal(np.asarray(va), np.asarray(vu))
    gp.close()


@pytest.mark.parametrize("D,N,M", [(1, 9, 3), (32, 61, 160), (8, 300, 1000), (16, 500, 4096)])
def test_value_only_small_evaluations_through_mapped_memory_are_bit_identical(ctx, oracle, D, N, M, monkeypatch):
    """sls_acq_eval without gradients on a small problem (what every DIRECT iteration of FindNextPoint issues, src/acquisition-
    function.cpp:155-165) reads its query points from and writes its values to page-locked memory the device maps -- one launch and
    one synchronisation instead of upload + launch + download.  Values only move: the bits of the copying path (SLS_EVAL_ZEROCOPY=0)."""
    monkeypatch.setenv("SLS_WAVE_PATH", "1")
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, M)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    for acq, h in ((0, 1.0), (1, 0.7)):
        v1 = gp.acq_eval(Xs, acq, h, want_grad=False)
        v1b = gp.acq_eval(Xs[:, :max(1, M // 3)], acq, h, want_grad=False)         # a smaller batch re-uses the block
        monkeypatch.setenv("SLS_EVAL_ZEROCOPY", "0")
        v0 = gp.acq_eval(Xs, acq, h, want_grad=False)
        monkeypatch.delenv("SLS_EVAL_ZEROCOPY")
        assert np.array_equal(v1, v0) and np.array_equal(v1b, v0[:max(1, M // 3)])
        vo = oracle.Regressor(X, y, theta, b, kernel=1).acq_eval_batch(Xs[:, :64], acq, h, want_grad=False)
        np.testing.assert_allclose(v1[:64], vo[:v1[:64].size], rtol=1e-6, atol=1e-9 * max(np.abs(vo).max(), 1e-30))
    gp.close()


@pytest.mark.parametrize("sigma_mode", [0, 1])
@pytest.mark.parametrize("D,N,n_local", [(1, 20, 40), (32, 61, 320), (32, 64, 60), (8, 65, 80), (16, 128, 50), (70, 100, 25)])
def test_single_start_cooperative_form_is_bit_identical(ctx, oracle, D, N, n_local, sigma_mode, monkeypatch):
    """A launch with ONE start (the local phase of the reference's DIRECT -> L-BFGS branch, src/acquisition-function.cpp:155-165) lets the
    four waves of its workgroup share every long sum of an evaluation (kernels_wave.hip, COOP) instead of shadowing each other.  Every
    sum is four fixed chains added as (c0 + c1) + (c2 + c3) whether one wave computes all four or four waves one each: the start
    must end with exactly the bits it gets as one of many starts of a larger launch, with the cooperative form switched off
    (SLS_WAVE_COOP=0), staged or not, in both sigma modes."""
    monkeypatch.setenv("SLS_WAVE_PATH", "1")
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, 9)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    gp.set_sigma_mode(sigma_mode)
    many = gp.acq_maximize(starts, n_local)                      # one wave per start
    for k in (0, 4):
        one = gp.acq_maximize(starts[:, k:k + 1], n_local)       # cooperative
        issued = gp.last_stats()["evals_issued"]
        assert one["y_stars"][0] == many["y_stars"][k] and np.array_equal(one["x_stars"][:, 0], many["x_stars"][:, k])
        monkeypatch.setenv("SLS_WAVE_COOP", "0")
        solo = gp.acq_maximize(starts[:, k:k + 1], n_local)      # one wave, three shadows
        assert solo["y_stars"][0] == one["y_stars"][0] and np.array_equal(solo["x_stars"], one["x_stars"])
        assert gp.last_stats()["evals_issued"] == issued
        monkeypatch.setenv("SLS_WAVE_STAGE", "0")
        monkeypatch.delenv("SLS_WAVE_COOP")
        unstaged = gp.acq_maximize(starts[:, k:k + 1], n_local)  # cooperative, K^-1 / X~ from global memory
        assert unstaged["y_stars"][0] == one["y_stars"][0] and np.array_equal(unstaged["x_stars"], one["x_stars"])
        monkeypatch.delenv("SLS_WAVE_STAGE")
    gp.close()


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("D,N,S", [(1, 20, 100), (32, 90, 10), (70, 300, 33), (128, 500, 9)])
def test_wave_path_matches_tiled_path_and_oracle(ctx, oracle, kernel, D, N, S, monkeypatch):
    """Small problems run one wavefront per start (kernels_wave.hip); forcing the tiled MFMA path on the same inputs and
    the oracle must give the same end points (summation order apart)."""
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    for acq in (0, 1):
        monkeypatch.setenv("SLS_WAVE_PATH", "1")
        rw = gp.acq_maximize(starts, 15, acq, 1.5)
        sw = gp.last_stats()
        monkeypatch.setenv("SLS_WAVE_PATH", "0")
        rt = gp.acq_maximize(starts, 15, acq, 1.5)
        st = gp.last_stats()
        # both paths count the evaluations of starts still moving: equal unless a start took another branch
        assert 0 < sw["evals_issued"] <= sw["evals_cap"] == S * 15
        assert abs(sw["evals_issued"] - st["evals_issued"]) <= 2 * 15, (sw, st)
        ro = oracle.Regressor(X, y, theta, b, kernel=kernel).acq_maximize(starts, 15, acq, 1.5, diag=True)
        assert_starts_agree(rw, ro, min_frac=0.9, max_divergent=2, label=f"wave D={D} N={N} S={S} kernel={kernel} acq={acq}")
        assert_starts_agree(rt, ro, min_frac=0.9, max_divergent=2, label=f"tiled D={D} N={N} S={S} kernel={kernel} acq={acq}")
        for other in (rt, ro):
            close(rw["value"], other["value"], rtol=RTOL)
            close(rw["x"], other["x"], rtol=RTOL, atol=1e-7)
        assert np.all((rw["x_stars"] >= 0) & (rw["x_stars"] <= 1))
    monkeypatch.delenv("SLS_WAVE_PATH")
    gp.close()


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("D,N,M,grow", [(6, 700, 300, 0), (16, 1500, 700, 0), (3, 126, 40, 5)])
def test_triangular_prediction_path(ctx, oracle, kernel, D, N, M, grow, monkeypatch):
    """Gradient-free calls (sls_gp_predict, sls_acq_eval without gradient) contract with the triangular L^-1
    (var_gemm_kernel, N^2 flops per point); they must agree with the K^-1 form, with the oracle, and stay valid after
    rank-1 growth of the factor (sls_gp_append_point extends L^-1 by a row)."""
    monkeypatch.setenv("SLS_WAVE_PATH", "0")
    X, y, theta, b = synth_problem(oracle, D, N + grow)
    Xs = synth_candidates(oracle, D, M)
    gp = sls().GP(ctx, X[:, :N], y[:N], theta, b, kernel)
    for i in range(N, N + grow):
        gp.append_point(X[:, i], y[i])
    ref = oracle.Regressor(X, y, theta, b, kernel=kernel)
    mu_o, sg_o = ref.predict_batch(Xs)
    ei_o = ref.acq_eval_batch(Xs, want_grad=False)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SLS_TRI_PREDICT", flag)
        mu, sg = gp.predict(Xs)
        ei = gp.acq_eval(Xs, want_grad=False)
        close(mu, mu_o, rtol=1e-6, atol=1e-8)
        close(sg, sg_o, rtol=1e-6, atol=1e-8)
        close(ei, ei_o, rtol=1e-6, atol=1e-9 * max(np.abs(ei_o).max(), 1e-30))
        out[flag] = (mu, sg, ei)
    close(out["1"][1], out["0"][1], rtol=1e-8, atol=1e-10)
    gp.close()


@pytest.mark.parametrize("seed", [46, 67, 184, 197, 226, 288, 312, 323, 324, 344, 405, 440])
def test_preference_handles_meet_the_flat_tolerance_on_ill_conditioned_cases(ctx, oracle, seed, path):
    """The twelve seeds of the 500-seed sweep (profiles/r03_random_sweep.json) whose sigma only met 1e-6 PLUS cond(K_y) eps a / 2 sigma:
    D <= 5, sigma down to 1e-3, cond(K_y) up to 1.5e8.  That bound belongs to the explicit inverse -- GaussianProcessRegressor's
    formula (src/gaussian-process-regressor.cpp:241-255).  PreferenceRegressor solves with the Cholesky factor
    (src/preference-regressor.cpp:299-313,323-330), which is accurate to ~1e-12 there; a handle in SLS_SIGMA_CHOLESKY_SOLVE mode
    (what host/preference-regressor.cpp sets) must therefore meet the FLAT 1e-6 on sigma on exactly these cases, on both paths,
    with and without gradients -- against the oracle's restatement of that class (reg_type = preference: LLT.solve per point)."""
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.integers(1, 40)); N = int(rng.integers(2, 400)); M = int(rng.integers(1, 300)); kernel = int(rng.integers(0, 2))
    X = rng.uniform(0, 1, (D, N))
    y = np.sin(X.sum(axis=0) * rng.uniform(1, 4)) + 0.05 * rng.normal(size=N)
    theta = np.concatenate([[rng.uniform(0.1, 2.0)], rng.uniform(0.2, 1.5, D) * np.sqrt(max(D, 4) / 4.0)])
    b = float(10 ** rng.uniform(-6, -1))
    Xs = rng.uniform(-0.1, 1.1, (D, M))[:, :64]
    gp = sls().GP(ctx, X, y, theta, b, kernel)
    gp.set_sigma_mode(sls().SIGMA_CHOLESKY_SOLVE)
    ref = oracle.Regressor(X, y, theta, b, kernel=kernel, reg_type=oracle.REG_PREF)
    sg_o = np.array([ref.predict_sigma(Xs[:, j]) for j in range(Xs.shape[1])])
    dsg_o = np.stack([ref.predict_sigma_derivative(Xs[:, j]) for j in range(Xs.shape[1])], axis=1)
    a_sig = 1e-7 * np.sqrt(theta[0])
    _, sg = gp.predict(Xs)                                   # gradient-free route (triangular contraction)
    assert _within(sg, sg_o, 1e-6, a_sig), np.max(np.abs(sg - sg_o) / np.maximum(sg_o, a_sig))
    _, dsg = gp.predict_grad(Xs)
    ok = np.isfinite(dsg_o)
    assert np.array_equal(np.isfinite(dsg), ok)
    # d sigma = -(1 / sigma) J K^-1 k with K^-1 k = L^-T (L^-1 k) on both paths (two triangular passes in the wavefront kernel; a
    # plain product V = K* L^-T followed by the acq_gemm tiles with (L^-1)^T in the place of K^-1 in the tiled pipeline): measured on
    # these cases <= 3.4e-6 of a column's largest component
    colmax = np.max(np.abs(np.where(ok, dsg_o, 0.0)), axis=0, keepdims=True) + np.zeros_like(dsg_o)
    assert np.all(np.abs(dsg - dsg_o)[ok] <= 2e-5 * colmax[ok] + 1e-12), np.max(np.abs(dsg - dsg_o)[ok] / np.maximum(colmax[ok], 1e-300))
    # the gradient route of the acquisition functions uses the same sigma: UCB value = mu + h sigma
    v, g = gp.acq_eval(Xs, 1, 0.7)
    mu_o = np.array([ref.predict_mu(Xs[:, j]) for j in range(Xs.shape[1])])
    assert _within(v - mu_o, 0.7 * sg_o, 1e-6, 0.7 * a_sig + 1e-6 * np.abs(mu_o).max()), np.max(np.abs(v - mu_o - 0.7 * sg_o))
    # ... and the default mode really is the other formula on these cases (otherwise this test would not test anything)
    gp.set_sigma_mode(sls().SIGMA_EXPLICIT_INVERSE)
    v0, _ = gp.acq_eval(Xs, 1, 0.7)
    from util import record
    record("pref_sigma", seed=int(seed), path=path, D=D, N=N, kernel=kernel, b=b, sigma_min=float(sg_o.min()),
           solve_mode_rel_err=float(np.max(np.abs(sg - sg_o) / np.maximum(sg_o, a_sig))),
           explicit_inverse_rel_err=float(np.max(np.abs((v0 - mu_o) / 0.7 - sg_o) / np.maximum(sg_o, a_sig))))
    gp.close()


@pytest.mark.parametrize("D,N", [(1, 12), (8, 90), (32, 128), (6, 300)])
def test_gp_map_objective_batch_matches_single_evaluations(ctx, oracle, D, N):
    """sls_gp_nll_batch (the B independent points of one DIRECT iteration of the GP MAP fit; src/gaussian-process-regressor.cpp:294
    evaluates them one by one).  N <= 128: ONE launch, one workgroup per point -- bit for bit the values of B single
    sls_gp_nll_grad calls.  N > 128 (round 4): several bordered factorisations per persistent launch (quad and log-det from the
    factor alone) -- bit for bit the values of the same call with one point at a time, and equal to the full evaluation (explicit
    inverse) to rounding.  A point whose K_y is not positive definite comes back as -inf instead of failing the batch (point 5 is
    as close to singular as kernel parameters get)."""
    X, y, _, _ = synth_problem(oracle, D, N)
    rng = np.random.default_rng(D * 1000 + N)
    B = 37
    xs = np.exp(rng.uniform(np.log(1e-3), np.log(5.0), (B, D + 2)))
    xs[:, 1] = np.exp(rng.uniform(np.log(1e-6), np.log(1e-1), B))
    xs[5] = np.concatenate([[1.0, 1e-300], np.full(D, 50.0)])      # K_y = all-ones + ~0: not positive definite in fp64
    for kernel in (0, 1):
        h = sls().Nll(ctx, X, kernel)
        vb = h.gp_objective_batch(y, xs)
        for k in range(B):
            try:
                v = h.gp_objective(y, xs[k], want_grad=False)
            except sls().SlsError:
                v = -np.inf
            if N <= 128:
                assert (vb[k] == v) or (np.isneginf(vb[k]) and np.isneginf(v)), (kernel, k, vb[k], v)
            else:
                v1 = h.gp_objective_batch(y, xs[k:k + 1])[0]
                assert (vb[k] == v1) or (np.isneginf(vb[k]) and np.isneginf(v1)), (kernel, k, vb[k], v1)
                # against the full evaluation: rounding, scaled by the conditioning of K_y = K_f + b I (b down to 1e-6 here)
                assert (np.isneginf(vb[k]) and np.isneginf(v)) or abs(vb[k] - v) <= 1e-9 * max(1.0, abs(v)), (kernel, k, vb[k], v)
        vo = np.array([oracle.gp_map_objective(kernel, X, y, xs[k])[0] for k in (0, 1, 2)])
        np.testing.assert_allclose(vb[:3], vo, rtol=1e-8)
        h.close()


def test_lbfgs_opts_struct_size_versions(ctx, oracle):
    """sls_lbfgs_opts is caller-allocated and has grown (ftol_rel / xtol_rel, round 5): struct_size says how much of it the caller's
    header knows.  A struct of the FIRST version's size (up to max_backtracks) is accepted and the members beyond it take the
    library's defaults (tolerances 0 = run to the cap: the same bits as no options at all); a size of 0 (a caller that never called
    sls_lbfgs_default_opts) or one larger than the library's own struct is refused with an error, not read past its end."""
    X, y, theta, b = synth_problem(oracle, 4, 60)
    starts = synth_candidates(oracle, 4, 16)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    L = sls().LbfgsOpts
    full = L(6, 1e-4, 0.5, 0.0, 20, 0.0, 0.0)
    assert full.struct_size == C.sizeof(L)
    r_none = gp.acq_maximize(starts, 12)
    r_full = gp.acq_maximize(starts, 12, opts=full)
    assert np.array_equal(r_none["y_stars"], r_full["y_stars"])
    old = L(6, 1e-4, 0.5, 0.0, 20, 123.0, 456.0)            # what sits behind the old struct's end must not be read
    old.struct_size = L.max_backtracks.offset + C.sizeof(C.c_int)
    r_old = gp.acq_maximize(starts, 12, opts=old)
    assert np.array_equal(r_none["y_stars"], r_old["y_stars"]) and np.array_equal(r_none["x_stars"], r_old["x_stars"])
    for bad in (0, C.sizeof(L) + 8, 4):
        o = L()
        o.struct_size = bad
        with pytest.raises(sls().SlsError, match="struct_size"):
            gp.acq_maximize(starts, 12, opts=o)
    gp.close()


@pytest.mark.parametrize("path,D,N,S", [("wave", 5, 40, 48), ("wave", 32, 61, 1), ("wave", 12, 200, 24), ("reg", 6, 300, 400), ("mem", 70, 256, 200)])
def test_relative_stopping_tests_of_nlopt_end_the_searches(ctx, oracle, path, D, N, S, monkeypatch):
    """sls_lbfgs_opts.ftol_rel / xtol_rel (round 5): NLopt's relative stopping tests (nlopt/src/util/stop.c: relstop) applied to every
    accepted step.  Every search of the reference runs through nloptutil::solve with both at 1e-6 (the host
    layer passes them), so a search ends with the first accepted step that changes the value or every coordinate by less than
    that fraction instead of polishing the 7th to 10th digit up to its evaluation cap.  The tests are discrete decisions like the
    Armijo test: a start whose step sits within rounding of the threshold may stop one step apart on the two sides, which moves
    its end value by about the tolerance itself -- the end values are held to 5e-6 (all of them), the chosen maximum likewise,
    the evaluation count must fall well below the cap, the one-wave-per-start / cooperative / register / memory forms must agree
    in every bit, and tolerances of 0 must reproduce the run without options."""
    monkeypatch.setenv("SLS_WAVE_PATH", "1" if path == "wave" else "0")
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, max(S, 2))[:, :S]
    n_local = 200
    gp = sls().GP(ctx, X, y, theta, b, 1)
    ref = oracle.Regressor(X, y, theta, b, kernel=1)
    opts = sls().LbfgsOpts(6, 1e-4, 0.5, 0.0, 20, 1e-6, 1e-6)
    r = gp.acq_maximize(starts, n_local, opts=opts)
    stats = gp.last_stats()
    ro = ref.acq_maximize(starts, n_local, ftol_rel=1e-6, xtol_rel=1e-6)
    scale = max(np.abs(ro["y_stars"]).max(), 1e-300)
    assert np.allclose(r["y_stars"], ro["y_stars"], rtol=5e-6, atol=1e-9 * scale), np.max(np.abs(r["y_stars"] - ro["y_stars"]) / scale)
    assert abs(r["value"] - ro["value"]) <= 5e-6 * abs(ro["value"]) + 1e-9 * scale
    assert np.all(r["x_stars"] >= 0.0) and np.all(r["x_stars"] <= 1.0)
    assert stats["evals_issued"] < 0.6 * stats["evals_cap"], stats
    # the same run in the other forms of the same path
    if path == "wave":
        for name in ("SLS_WAVE_COOP", "SLS_WAVE_STAGE"):
            with env_switch(name, 0):
                r2 = gp.acq_maximize(starts, n_local, opts=opts)
            assert np.array_equal(r2["y_stars"], r["y_stars"]) and np.array_equal(r2["x_stars"], r["x_stars"])
    elif path == "reg":
        with env_switch("SLS_LBFGS_REG", 0):
            r2 = gp.acq_maximize(starts, n_local, opts=opts)
        assert np.array_equal(r2["y_stars"], r["y_stars"]) and np.array_equal(r2["x_stars"], r["x_stars"])
    with env_switch("SLS_COMPACT", 0):
        r3 = gp.acq_maximize(starts, n_local, opts=opts)
    assert np.array_equal(r3["y_stars"], r["y_stars"])
    # tolerances 0 = no options
    r0 = gp.acq_maximize(starts, 30, opts=sls().LbfgsOpts(6, 1e-4, 0.5, 0.0, 20, 0.0, 0.0))
    rn = gp.acq_maximize(starts, 30)
    assert np.array_equal(r0["y_stars"], rn["y_stars"]) and np.array_equal(r0["x_stars"], rn["x_stars"])
    # ... and what stopping early costs.  NLopt's tests look at ONE accepted step: a single start may stop on a short step far from
    # its optimum (recorded: worst_loss_rel up to 0.66 of the scale on the 400-start case).  A multi-start MAXIMUM absorbs most of
    # that, not all: MEASURED on these shapes, the chosen maximum ends up to 1.0e-3 of its value below the capped run's (D = 70,
    # N = 256, 200 starts: 0.266195 against 0.266453); the bound asserted is 5e-3, the loss is recorded.  That is the price of
    # nloptutil::solve's defaults, which the reference pays as well (recollection).  Where a single search
    # picks the answer, the DIRECT -> L-BFGS branch at C3's shapes, the measured loss is 5e-6 (tests/test_gpu_early_stop.py).  No
    # start may end above its own continuation.
    rc = gp.acq_maximize(starts, n_local)
    assert np.all(rc["y_stars"] >= r["y_stars"] - 1e-12 * scale)
    if S > 1:
        assert rc["value"] - r["value"] <= 5e-3 * scale, (rc["value"], r["value"])
    record("nlopt_tolerances", path=path, D=D, N=N, S=S, evals_issued=int(stats["evals_issued"]), evals_cap=int(stats["evals_cap"]),
           worst_loss_rel=float(np.max((rc["y_stars"] - r["y_stars"]) / scale)), max_loss_rel=float((rc["value"] - r["value"]) / scale))
    gp.close()
