"""Multi-GPU maximisation behind the C ABI (sls_multi_* / sls_comm_*, include/sls_hip.h) on the 1-GPU box.

RCCL allows one rank per GPU, so the sharding logic is exercised with logical shards on device 0 (replicated fit, the start
set split into contiguous slices with global index offsets, per-shard winners merged by first maximum on the host), and the
RCCL code path itself (dlopen, communicator, ncclAllGather on the context's stream, merge of the gathered records) with
one-rank communicators: ncclCommInitAll over [0] and ncclCommInitRank with world = 1."""
import numpy as np
import pytest

from util import sls, synth_candidates, synth_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0]])
def test_sharded_maximisation_reproduces_single_device_winner(oracle, devices):
    m = sls()
    D, N, S, n_local = 6, 200, 333, 12
    X, y, theta, b = synth_problem(oracle, D, N)
    starts = synth_candidates(oracle, D, S)
    ctx = m.Context(0)
    gp = m.GP(ctx, X, y, theta, b, 1)
    one = gp.acq_maximize(starts, n_local)
    one_issued = gp.last_stats()["evals_issued"]
    multi = m.Multi(devices)
    if len(devices) == 1:
        assert multi.exchange == "ncclAllGather", multi.exchange      # RCCL loaded and initialised inside the library
    else:
        assert multi.exchange.startswith("host merge"), multi.exchange
    mgp = m.MultiGP(multi, X, y, theta, b, 1)
    for _ in range(2):                                                 # second call re-uses buffers / communicators
        r = mgp.acq_maximize(starts, n_local)
        assert r["index"] == one["index"] and r["value"] == one["value"] and np.array_equal(r["x"], one["x"])
        assert r["evals_issued"] == one_issued                         # the same starts retire at the same evaluation
    mgp.close(); multi.close(); gp.close(); ctx.close()


@pytest.mark.parametrize("M", [1, 2, 4096])
def test_sharded_predict_is_bit_identical(oracle, M):
    """BASELINE config C2's predict over several devices (candidate columns shard with no collective): three
    logical shards on the one GPU, against the single-device call -- also with fewer points than shards."""
    m = sls()
    D, N = 16, 2048 if M > 2 else 300
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, M)
    ctx = m.Context(0)
    gp = m.GP(ctx, X, y, theta, b, 0)
    mu1, s1 = gp.predict(Xs)
    multi = m.Multi([0, 0, 0])
    mgp = m.MultiGP(multi, X, y, theta, b, 0)
    mu3, s3 = mgp.predict(Xs)
    assert np.array_equal(mu1, mu3) and np.array_equal(s1, s3)
    mgp.close(); multi.close(); gp.close(); ctx.close()


def test_more_shards_than_starts_and_ties():
    """Two starts over three shards (one shard idle), identical starts (a tie: the lowest global index wins)."""
    m = sls()
    rng = np.random.default_rng(5)
    X = rng.uniform(0, 1, (2, 30)); y = np.sin(4 * X[0]) + X[1]
    theta = np.array([0.5, 0.3, 0.3])
    x0 = rng.uniform(0, 1, (2, 1))
    starts = np.concatenate([x0, x0], axis=1)
    multi = m.Multi([0, 0, 0])
    mgp = m.MultiGP(multi, X, y, theta, 0.01, 1)
    r = mgp.acq_maximize(starts, 6)
    assert r["index"] == 0
    ctx = m.Context(0)
    gp = m.GP(ctx, X, y, theta, 0.01, 1)
    one = gp.acq_maximize(starts, 6)
    assert one["index"] == 0 and one["value"] == r["value"]
    mgp.close(); multi.close(); gp.close(); ctx.close()


def test_rccl_communicator_single_rank_allgather():
    """sls_comm_*: the one-process-per-GPU exchange bench.py uses for --gpus N, here with world = 1."""
    m = sls()
    ctx = m.Context(0)
    uid = m.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = m.Comm(ctx, uid, 0, 1)
    x = np.linspace(0.1, 0.9, 64)
    for k in range(3):
        v, i, xo = comm.allgather_best(1.25 + k, 4242 + k, x)
        assert v == 1.25 + k and i == 4242 + k and np.array_equal(xo, x)
    comm.close(); ctx.close()


@pytest.mark.parametrize("N", [1500, 2700])      # static teams + three-workgroup chain / dynamic pools per XCD (round 6)
def test_concurrent_contexts_on_one_device_do_not_fall_back(oracle, N):
    """Two host threads, one context each, on the SAME GPU, fitting at the same time (what sls_multi does with a repeated
    device, and what sls_hip.h recommends for multi-threaded callers).  The single-launch Cholesky needs every workgroup resident
    at once, so two of them in flight would starve each other into their bounded-wait fallback; launches are serialised per
    device instead: same bits as a lone fit, and no fallback recorded on either context."""
    import threading
    m = sls()
    D = 6
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, 64)
    c0 = m.Context(0)
    g = m.GP(c0, X, y, theta, b, 1)
    ref = g.predict(Xs)
    g.close()
    out, ctxs = {}, [m.Context(0), m.Context(0)]

    def work(i):
        res = []
        for _ in range(6):
            gp = m.GP(ctxs[i], X, y, theta, b, 1)
            res.append(gp.predict(Xs))
            gp.close()
        out[i] = res

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    for i in range(2):
        for mu, sg in out[i]:
            assert np.array_equal(mu, ref[0]) and np.array_equal(sg, ref[1])
        assert ctxs[i].prof_get("potrf_fallbacks")[1] == 0
        ctxs[i].close()
    c0.close()


def test_single_launch_cholesky_gives_up_falls_back_and_rearms(oracle, monkeypatch):
    """The failure path of the single-launch Cholesky, forced: SLS_POTRF_TIMEOUT_TICKS=1 makes every device-side wait of the
    dataflow kernel expire (what happens when its workgroups cannot all be resident: a second process on the GPU, a profiler).
    The fit must still come out right -- recomputed on the multi-launch schedule --, the context counts ONE fallback, stays on
    the multi-launch schedule for the next 16 factorisations without further aborts, then goes back to the single-launch form."""
    m = sls()
    D, N = 6, 1000
    X, y, theta, b = synth_problem(oracle, D, N)
    Xs = synth_candidates(oracle, D, 64)
    c0 = m.Context(0)
    g = m.GP(c0, X, y, theta, b, 1)
    mu0, s0 = g.predict(Xs)
    g.close()
    c = m.Context(0)
    monkeypatch.setenv("SLS_POTRF_TIMEOUT_TICKS", "1")
    g = m.GP(c, X, y, theta, b, 1)
    mu1, s1 = g.predict(Xs)
    g.close()
    assert c.prof_get("potrf_fallbacks")[1] == 1
    np.testing.assert_allclose(mu1, mu0, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(s1, s0, rtol=1e-9, atol=1e-12)
    for _ in range(14):                                   # multi-launch fits (16 factorisations, the recomputation included): no further fallback
        g = m.GP(c, X, y, theta, b, 1)
        g.close()
    assert c.prof_get("potrf_fallbacks")[1] == 1
    monkeypatch.delenv("SLS_POTRF_TIMEOUT_TICKS")
    for _ in range(3):                                    # re-armed: the single-launch form again, now with its normal waits
        g = m.GP(c, X, y, theta, b, 1)
        mu2, s2 = g.predict(Xs)
        g.close()
    assert c.prof_get("potrf_fallbacks")[1] == 1
    assert np.array_equal(mu2, mu0) and np.array_equal(s2, s0)
    monkeypatch.setenv("SLS_POTRF_TIMEOUT_TICKS", "1")      # and it gives up again when the condition returns
    g = m.GP(c, X, y, theta, b, 1)
    g.close()
    assert c.prof_get("potrf_fallbacks")[1] == 2
    c.close(); c0.close()


def test_map_evaluation_survives_a_cholesky_that_gives_up(oracle, monkeypatch):
    """The same forced expiry inside sls_gp_nll_grad: the evaluation enqueues factorisation and gradient in one go and reads the
    factorisation's status back with the results (capi_map.hip); when the status says "gave up" the WHOLE evaluation is run
    once more on the multi-launch schedule -- value and gradient must be those of an undisturbed context."""
    m = sls()
    D, N = 5, 700
    X, y, _, _ = synth_problem(oracle, D, N)
    x = np.concatenate([[0.6, 0.01], np.linspace(0.4, 0.9, D)])
    c0 = m.Context(0)
    h0 = m.Nll(c0, X, 1)
    v0, g0 = h0.gp_objective(y, x)
    h0.close()
    c = m.Context(0)
    h = m.Nll(c, X, 1)
    monkeypatch.setenv("SLS_POTRF_TIMEOUT_TICKS", "1")
    v1, g1 = h.gp_objective(y, x)
    assert c.prof_get("potrf_fallbacks")[1] == 1
    np.testing.assert_allclose(v1, v0, rtol=1e-10)
    np.testing.assert_allclose(g1, g0, rtol=1e-7, atol=1e-9 * np.abs(g0).max())
    v2, g2 = h.gp_objective(y, x * 1.01)                    # the next evaluation: multi-launch from the start, no new fallback
    assert c.prof_get("potrf_fallbacks")[1] == 1 and np.isfinite(v2) and np.all(np.isfinite(g2))
    h.close(); c.close(); c0.close()


@pytest.mark.parametrize("N", [90, 300])
def test_map_objective_batch_over_logical_shards_is_bit_identical(oracle, N):
    """sls_multi_gp_nll_batch: the points of a DIRECT iteration dealt round-robin over the devices (three logical shards on GPU 0
    here) give bit for bit the values of sls_gp_nll_batch on one device -- the part of a MAP fit (BASELINE config 5) that shards."""
    m = sls()
    D = 8
    X, y, _, _ = synth_problem(oracle, D, N)
    rng = np.random.default_rng(N)
    xs = np.exp(rng.uniform(np.log(1e-2), np.log(3.0), (11, D + 2)))
    xs[:, 1] = np.exp(rng.uniform(np.log(1e-5), np.log(1e-2), 11))
    c = m.Context(0)
    h = m.Nll(c, X, 1)
    one = h.gp_objective_batch(y, xs)
    h.close()
    mg = m.Multi([0, 0, 0])
    mh = m.MultiNll(mg, X, 1)
    three = mh.gp_objective_batch(y, xs)
    mh.close()
    mg.close()
    c.close()
    assert np.array_equal(one, three)


def test_handles_may_outlive_their_context(oracle):
    """A garbage-collected binding destroys context and handles in no particular order (interpreter exit: the weak references
    Context.close walks are already dead): sls_ctx_destroy with live handles only marks the context, the handles keep working, and
    the last one to be destroyed frees it -- before, the handle's destructor locked a freed mutex."""
    m = sls()
    lib = m.lib()
    X, y, theta, b = synth_problem(oracle, 4, 200)
    Xs = synth_candidates(oracle, 4, 16)
    c = m.Context(0)
    g = m.GP(c, X, y, theta, b, 1)
    h = m.Nll(c, X, 1)
    mu0, s0 = g.predict(Xs)
    assert lib.sls_ctx_destroy(c.h) == 0          # the context goes first
    c.h = None
    mu1, s1 = g.predict(Xs)                          # ... and its handles still work
    assert np.array_equal(mu0, mu1) and np.array_equal(s0, s1)
    v, _ = h.gp_objective(y, np.concatenate([[0.5, 0.01], np.full(4, 0.5)]))
    assert np.isfinite(v)
    g.close()
    h.close()                                        # the last handle frees the context
    c2 = m.Context(0)                                # and the device is still usable
    g2 = m.GP(c2, X, y, theta, b, 1)
    assert np.array_equal(g2.predict(Xs)[0], mu0)
    g2.close(); c2.close()
