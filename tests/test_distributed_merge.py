"""N > 1 path on CPU: two gloo ranks shard one start set, maximise their slice (with the ORACLE standing in for the
device kernels -- allowed in tests only), exchange (value, global index, x) with ONE all-gather and must reproduce the
single-rank answer, including the first-maximum tie-break."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _problem():
    from oracle import oracle_py as orc
    from util import synth_candidates, synth_problem
    X, y, theta, b = synth_problem(orc, 3, 40)
    starts = synth_candidates(orc, 3, 37)      # 37 starts: uneven split over 2 ranks
    starts[:, 30] = starts[:, 5]               # duplicated start -> exact tie in value, lower index must win
    return orc, X, y, theta, b, starts


def _worker(rank, world, port, q):
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sls = importlib.import_module("sequential-line-search_amd")
    orc, X, y, theta, b, starts = _problem()
    ref = orc.Regressor(X, y, theta, b, kernel=1)
    lo, hi = sls.shard_range(starts.shape[1], rank, world)
    r = ref.acq_maximize(starts[:, lo:hi], 12, n_threads=1)
    v, i, x = sls.exchange_best(r["value"], lo + r["index"], r["x"])
    q.put((rank, lo, hi, v, i, x.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shard_and_merge_matches_single_rank():
    import importlib
    sls = importlib.import_module("sequential-line-search_amd")
    assert [sls.shard_range(37, r, 2) for r in range(2)] == [(0, 19), (19, 37)]
    assert [sls.shard_range(65536, r, 8)[1] - sls.shard_range(65536, r, 8)[0] for r in range(8)] == [8192] * 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    orc, X, y, theta, b, starts = _problem()
    full = orc.Regressor(X, y, theta, b, kernel=1).acq_maximize(starts, 12, n_threads=1)
    for rank, lo, hi, v, i, x in res:
        assert i == full["index"], (rank, i, full["index"])
        assert v == full["value"]
        assert np.array_equal(np.array(x), full["x"])
    # the duplicated start must never win over its lower-index twin
    assert full["index"] != 30
