"""Concurrent const evaluations and the predictor generation (round 5).

Reference: the multi-start loop calls Predict* of ONE const regressor from hardware_concurrency worker threads
(src/acquisition-function.cpp:125-144).  Here a call with up to 64 points on a small handle borrows a stream + mapped block of the
context (csrc/capi.hip: eval_in_slot) under a shared lock on the fitted state, so host threads overlap; everything that changes the
state holds that lock exclusively."""
import threading

import numpy as np
import pytest

from util import env_switch, sls, synth_candidates, synth_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = sls().Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("sigma_mode", [0, 1])
def test_threads_on_one_handle_return_the_sequential_bits(ctx, oracle, sigma_mode):
    D, N, T, M = 6, 90, 8, 120
    X, y, theta, b = synth_problem(oracle, D, N)
    gp = sls().GP(ctx, X, y, theta, b, 1)
    gp.set_sigma_mode(sigma_mode)
    Q = synth_candidates(oracle, D, M)
    seq = [(gp.predict(Q[:, i:i + 1]), gp.predict_grad(Q[:, i:i + 1]), gp.acq_eval(Q[:, i:i + 1])) for i in range(M)]
    with env_switch("SLS_EVAL_SLOTS", 0):           # the locked path: same kernel, same bits
        locked = [(gp.predict(Q[:, i:i + 1]), gp.predict_grad(Q[:, i:i + 1]), gp.acq_eval(Q[:, i:i + 1])) for i in range(0, M, 7)]
    for k, i in enumerate(range(0, M, 7)):
        for a, b_ in zip(seq[i], locked[k]):
            assert all(np.array_equal(u, v) for u, v in zip(a, b_))
    out = [None] * T
    err = []

    def worker(t):
        try:
            out[t] = [(gp.predict(Q[:, i:i + 1]), gp.predict_grad(Q[:, i:i + 1]), gp.acq_eval(Q[:, i:i + 1])) for i in range(M)]
        except Exception as e:      # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for t in range(T):
        for i in range(M):
            for a, b_ in zip(seq[i], out[t][i]):
                assert all(np.array_equal(u, v) for u, v in zip(a, b_)), (t, i)
    # against the oracle, once
    ref = oracle.Regressor(X, y, theta, b, kernel=1)
    mu_o, sg_o = ref.predict_batch(Q)
    np.testing.assert_allclose(np.concatenate([s[0][0] for s in seq]), mu_o, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.concatenate([s[0][1] for s in seq]), sg_o, rtol=1e-6, atol=1e-9)
    gp.close()


def test_readers_and_a_writer_interleave(ctx, oracle):
    """Threads predicting while another appends points: every answer belongs to SOME consistent state (before or after an append,
    never a mixture), and the final state equals a refit on the grown data."""
    D, N = 4, 60
    X, y, theta, b = synth_problem(oracle, D, N)
    extra = synth_candidates(oracle, D, 5, seed=99)
    ys = [0.3, 0.1, 0.25, 0.2, 0.15]
    gp = sls().GP(ctx, X, y, theta, b, 1)
    q = synth_candidates(oracle, D, 1, seed=5)
    states = [gp.predict(q)[0][0]]
    g2 = sls().GP(ctx, X, y, theta, b, 1)
    for k in range(5):
        g2.append_point(extra[:, k], ys[k])
        states.append(g2.predict(q)[0][0])
    g2.close()
    seen, stop = [], threading.Event()

    def reader():
        while not stop.is_set():
            seen.append(gp.predict(q)[0][0])
    th = [threading.Thread(target=reader) for _ in range(4)]
    for t in th:
        t.start()
    gens = [gp.generation()]
    for k in range(5):
        gp.append_point(extra[:, k], ys[k])
        gens.append(gp.generation())
    stop.set()
    for t in th:
        t.join()
    assert len(set(gens)) == 6 and gens == sorted(gens)
    assert len(seen) > 0 and all(any(v == s for s in states) for v in seen), "a reader saw a state no append ever produced"
    assert gp.predict(q)[0][0] == states[-1]
    gp.close()


def test_generation_changes_with_the_predictor(ctx, oracle):
    X, y, theta, b = synth_problem(oracle, 3, 40)
    a, c = sls().GP(ctx, X, y, theta, b, 1), sls().GP(ctx, X, y, theta, b, 1)
    ga, gc = a.generation(), c.generation()
    assert ga != gc and ga > 0
    a.set_sigma_mode(1)
    assert a.generation() > ga and c.generation() == gc
    g1 = a.generation()
    a.set_sigma_mode(1)                             # no change: same predictor
    assert a.generation() == g1
    a.close(); c.close()
    d = sls().GP(ctx, X, y, theta, b, 1)            # possibly at a recycled address: still a new number
    assert d.generation() > max(g1, gc)
    d.close()
