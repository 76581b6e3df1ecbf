"""CPU model check of the dataflow Cholesky schedule (csrc/kernels_chol.hip: potrf_dataflow_kernel) with the discrete-event
model of tools/potrf_dataflow_sim.py: the same ownership, the same fixed chunking rule, the same "first ready task among the
next 16 unfinished tiles in column order" policy, one chain.  Task durations are randomised (seeded), so that owners finish in
every relative order: the schedule must always run to completion -- no circular wait, also with the 16-tile window and with
far fewer workers than tiles -- and the chain must take every step exactly once."""
import importlib.util
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("potrf_dataflow_sim", os.path.join(ROOT, "tools", "potrf_dataflow_sim.py"))
sim = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sim)


@pytest.mark.parametrize("nb", [3, 4, 9, 16, 33, 64])
@pytest.mark.parametrize("nbo,near", [(1, 0), (2, 0), (4, 3), (8, 0), (8, 4)])
def test_schedule_completes_for_any_task_timing(nb, nbo, near):
    rng = random.Random(1000 * nb + 10 * nbo + near)
    for trial in range(3):
        W = rng.choice([5, 17, 64, 255])
        PR = rng.choice([1, 2, 6, 12])
        PR = min(PR, W)
        r = sim.simulate(nb, PR=PR, nbo=nbo, W=W, near=near,
                         chain_step=(rng.uniform(5, 30), rng.uniform(5, 30), rng.uniform(10, 60)),
                         t_gemm=rng.uniform(2, 40), t_rmw=rng.uniform(0.5, 10), t_sched=rng.uniform(0.5, 8))
        assert r["total_us"] < float("inf") and r["total_us"] > 0
        assert r["steps_done"] == nb - 1, r                   # every chain step ran
        assert r["tiles_unfinished"] == 0, r                  # every owned tile got all its updates and its panel solve


def test_chunks_cover_every_step_once():
    """The chunking rule (df_chunk_end in the kernel, `chunks` in the model): the chunks of tile (i, k) partition the steps the
    owner applies, 0 .. k-1 (k-2 for a diagonal tile: the chain applies the last one), in order, whatever nbo / near."""
    for nb in (5, 16, 40):
        for nbo in (1, 2, 3, 4, 8):
            for near in (0, 2, 8):
                for k in range(nb):
                    for i in (k, min(k + 1, nb - 1), nb - 1):
                        if i == 0:
                            continue
                        ch = sim.chunks_of(i, k, nbo, near)
                        target = k - 1 if i == k else k
                        flat = [j for (j0, j1) in ch for j in range(j0, j1)]
                        assert flat == list(range(max(target, 0))), (nb, nbo, near, i, k, ch)
                        assert all(j1 - j0 <= nbo for j0, j1 in ch)
