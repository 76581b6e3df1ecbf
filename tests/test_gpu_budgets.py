"""Time budgets of the BASELINE configurations that are NOT the bench line, asserted where the driver's own GPU test run sees them
(GPUTEST_rNN.json), not only in the builder's profiles/.  The budgets are 1.25-1.3x the values measured on an MI355X in the round named per test (round 5 tightened them from 1.5-2x,
and made the C3 tests assert ONE fused launch per fit): they catch a path that silently fell back to a slower schedule, with
room for box-to-box noise only.  The reference's
only timing artefact is the wall time of SubmitFeedbackData (demos/sequential_line_search_nd/main.cpp:86-91,114)."""
import os
import re
import subprocess
import time

import numpy as np
import pytest

from util import env_switch, record, sls, synth_candidates, synth_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "sequential-line-search_amd", "bin")


@pytest.fixture(scope="module")
def ctx():
    c = sls().Context(0)
    yield c
    c.close()


def best_of(f, reps, sync):
    f(); sync()
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        f(); sync()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def test_c2_fit_and_predict_budget(ctx, oracle):
    """C2: N = 2048, D = 16, ARD-SE: Gram + Cholesky + K^-1 + alpha, wall time including the upload of X / y; 4096-point predict
    including the PCIe transfers.  Measured: 1.0 ms / 0.52 ms (first session of round 4, three launches for factor and inverse
    on the unstreamed chain: 1.43 ms)."""
    D, N, M = 16, 2048, 4096
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, M)
    m = sls()
    fit = best_of(lambda: m.GP(ctx, X, y, theta, b, 0).close(), 5, ctx.synchronize)
    gp = m.GP(ctx, X, y, theta, b, 0)
    pred = best_of(lambda: gp.predict(Xs), 5, ctx.synchronize)
    gp.close()
    record("budget", config="C2", fit_ms=fit, predict_ms=pred)
    # bounds at 1.25x the measured values (round 6: 0.83-0.84 ms and 0.38 ms): a regression to the round-3 chain or to pageable copies fails
    assert fit <= 1.05, fit
    assert pred <= 0.5, pred


def test_factor_and_inverse_device_time_budget(oracle):
    """Factor + L^-1 + K^-1 of a fit, device time from the library's own HIP-event scopes (sls_prof_get): ONE launch up to
    N = 4096 (kernels_chol.hip: streamed chain + fused inverse).  Measured: 0.74-0.78 ms at N = 2048 (first session of round 4:
    potrf 0.79 + trtri 0.27 + lauum 0.14 = 1.2 ms), 2.16-2.2 ms at N = 4096 (2.62)."""
    m = sls()
    for N, budget in ((2048, 0.9), (4096, 2.4)):       # 1.25x the measured 0.70-0.72 / 1.89 ms in this test (round 6: dynamic pools at 4096; round 5: 2.15-2.2)
        X, y, theta, b = synth_problem(oracle, 16, N)
        c = m.Context(0)
        m.GP(c, X, y, theta, b, 0).close()            # code objects, buffers
        c.prof_enable(True); c.prof_reset()
        for _ in range(3):
            m.GP(c, X, y, theta, b, 0).close()
        ms, launches = c.prof_get("potri")
        # ONE fused launch per fit: a silent fall-back to potrf + trtri + lauum shows up under another scope name and fails here
        assert launches == 3 and c.prof_get("potrf")[1] == 0 and c.prof_get("potrf+trtri+lauum")[1] == 0 and c.prof_get("potrf_fallbacks")[1] == 0
        record("budget", config="factor+inverse", N=N, device_ms=ms / 3)
        assert ms / 3 <= budget, (N, ms / 3)
        c.close()


# ~1.3x the measured means (MAP on: 2.96-3.0 ms, fixed: 1.44-1.46 ms) with the local searches stopped by nloptutil::solve's relative
# tolerances (the host layer's default; to their caps: 3.75-4.2 / 1.9-2.45 ms, depending on how many local phases ran into the cap)
@pytest.mark.parametrize("use_map,budget_ms", [(1, 3.75), (0, 1.75)])   # 1.25x the measured 2.97-3.0 / 1.39-1.43 ms (rounds 5 and 6)
def test_c3_submit_feedback_budget(use_map, budget_ms):
    """C3: sequential_line_search_nd, D = 32, 30 iterations: wall time of SubmitFeedbackData (preference MAP fit on the device +
    DIRECT -> L-BFGS acquisition maximisation), steady state (the first submit carries the one-off initialisation).
    use_map = 1: the reference's own setting (demos/sequential_line_search_nd/main.cpp:21, the constructor's default): the kernel
    hyper-parameters are part of the fit, every evaluation rebuilds and factors K.  use_map = 0: fixed hyper-parameters (K cached,
    src/preference-regressor.cpp:363-371): measured 2.5 ms in round 4 (round 3: 7.6 ms; the oracle on one host core: 6 ms mean)."""
    p = subprocess.run([os.path.join(BIN, "sequential_line_search_nd"), "32", "30", "1", str(use_map)], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    ms = [float(v) for v in re.findall(r" ms ([-\d.e]+)", p.stdout)]
    assert len(ms) == 30
    steady = float(np.mean(ms[1:]))
    record("budget", config="C3", use_map_hyperparams=bool(use_map), ms_per_submit_steady=steady, ms_per_submit_median=float(np.median(ms)),
           ms_max=float(np.max(ms[1:])), ms_last=ms[-1])
    assert steady <= budget_ms, steady


def test_c5_evaluation_and_batch_budget(ctx, oracle):
    """C5: Matern-5/2 MAP objective + gradient at N = 4096, D = 128 (measured 2.57-2.65 ms once the evaluation made no copy calls and its
    small launches were merged; before that 2.75-2.9; first session of round 4: 3.4), and the
    value-only evaluations of a DIRECT iteration: eight parameter sets through sls_gp_nll_batch (concurrent bordered
    factorisations, measured 6.4-6.6 ms) against one full evaluation after the other (SLS_NLL_BATCH=0, 19.8 ms): at least 1.8x."""
    D, N = 128, 4096
    X, y, theta, b = synth_problem(oracle, D, N)
    h = sls().Nll(ctx, X, 1)
    x = np.concatenate([[0.5, 0.005], np.full(D, theta[1])])
    k = [0]

    def ev():
        k[0] += 1
        xx = x.copy(); xx[2] *= 1 + 1e-3 * k[0]
        h.gp_objective(y, xx)
    ms_eval = best_of(ev, 4, ctx.synchronize)
    xs = np.tile(x, (8, 1)); xs[:, 2] *= 1 + 1e-3 * np.arange(8)
    ms_batch = best_of(lambda: h.gp_objective_batch(y, xs), 3, ctx.synchronize)
    with env_switch("SLS_NLL_BATCH", 0):
        ms_seq = best_of(lambda: h.gp_objective_batch(y, xs), 1, ctx.synchronize)
    h.close()
    record("budget", config="C5", ms_per_evaluation=ms_eval, batch8_ms=ms_batch, sequential8_ms=ms_seq, speedup=ms_seq / ms_batch)
    assert ms_eval <= 2.8, ms_eval     # 1.25x the measured 2.1-2.25 ms (round 6; round 5: 2.5-2.6)
    assert ms_seq / ms_batch >= 1.8, (ms_seq, ms_batch)


def test_c1_demo_budget():
    """C1: bayesian_optimization_1d, 20 iterations.  The run is process start (HIP initialisation + code objects: 0.19-0.26 s for
    a process that launches one trivial kernel) plus ~4 ms per iteration; measured 0.28-0.40 s."""
    t0 = time.perf_counter()
    p = subprocess.run([os.path.join(BIN, "bayesian_optimization_1d"), "1", "20", "1"], capture_output=True, text=True, timeout=300)
    wall = time.perf_counter() - t0
    assert p.returncode == 0, p.stderr[-2000:]
    record("budget", config="C1", wall_s=wall)
    assert wall <= 1.5, wall
