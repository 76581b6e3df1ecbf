"""bench.py contract and its N>1 code path, on the 1-GPU box.

Two ranks are launched through torch.distributed.run exactly as the driver does, but share GPU 0 and exchange over gloo
(RCCL refuses two ranks on one device). Everything else -- sharding of the starts, per-rank maximisation with the global
start offset, the single exchange, the first-maximum merge, max-over-ranks timing, the JSON line -- is the bench's own code.
The merged winner must equal the single-rank answer over all starts.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# long option names: torch.distributed.run would prefix-match "--n" / "--d" against its own options
SMALL = ["--num-train", "640", "--dims", "8", "--starts", "3000", "--n-local", "8", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-traffic"]
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_contract_and_two_rank_merge():
    one = run([sys.executable, "bench.py", "--gpus", "1"] + SMALL)
    for k in REQUIRED:
        assert k in one, k
    assert one["n_gpus"] == 1 and one["dtype"] == "f64" and one["scaling"] == "strong" and one["vs_baseline"] is None
    assert one["unit"] == "candidate-evals/s" and one["value"] > 0 and one["higher_is_better"] is True
    r = one["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "workload" in one["config"]
    c = one["config"]
    assert 0 < c["evals_issued_per_step"] <= c["evals_cap_per_step"] == 3000 * 8
    assert abs(one["value"] - c["evals_issued_per_step"] / (one["ms_per_step"] * 1e-3)) < 1e-6 * one["value"]
    for name, st in one["stage_rooflines"].items():          # a fraction above 1 is an accounting error, not evidence
        assert 0 < st["frac"] < 1, (name, st)
    assert "fit_pipeline" in one["stage_rooflines"] and r["traffic"] is None and "skipped" in r["traffic_detail"]   # --no-traffic here
    assert one["potrf_fallbacks"] == 0

    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", "29617", "bench.py", "--gpus", "2", "--backend", "gloo", "--same-device"] + SMALL)
    assert two["n_gpus"] == 2 and two["config"]["starts_per_gpu"] == 1500 and "test_mode" in two["config"]
    assert two["result"]["best_index"] == one["result"]["best_index"]
    assert two["result"]["best_value"] == one["result"]["best_value"]
    np.testing.assert_array_equal(two["result"]["best_x"], one["result"]["best_x"])


@pytest.mark.gpu
def test_bench_gpus_n_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2 ...` invoked PLAINLY (no torch.distributed.run around it, short option names and all) must
    not die: it starts the two ranks itself and prints the one JSON line, with the single-rank winner bit for bit
    (multi-start loop of src/acquisition-function.cpp:121-153 sharded over ranks)."""
    short = ["--n", "640", "--d", "8", "--starts", "3000", "--n-local", "8", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-traffic"]
    one = run([sys.executable, "bench.py", "--gpus", "1"] + short)
    two = run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--same-device"] + short)
    assert two["n_gpus"] == 2 and two["config"]["starts_per_gpu"] == 1500
    assert "started by bench.py itself" in two["config"]["launcher"] and one["config"]["launcher"] == "single process"
    assert two["result"]["best_index"] == one["result"]["best_index"]
    assert two["result"]["best_value"] == one["result"]["best_value"]
    np.testing.assert_array_equal(two["result"]["best_x"], one["result"]["best_x"])


@pytest.mark.gpu
def test_eight_rank_dry_run_of_the_full_shard_shape():
    """What the first 8-GPU lease will run, rehearsed on one GPU: 8 ranks through torch.distributed.run, each with the 8 192-start
    shard of the full N = 8192, D = 64 workload (its chunk shape, tail split and global start offsets; three evaluations per start to
    keep it short), the exchange over the rendezvous group.  The merged winner must be the single-rank winner over all 65 536
    starts, and the one JSON line must carry what makes a scaling curve diagnosable: per-rank step time, per-rank acq_gemm
    fraction, exchange time and the max - min skew (multi-start loop of src/acquisition-function.cpp:121-153)."""
    full = ["--num-train", "8192", "--dims", "64", "--starts", "65536", "--n-local", "3", "--steps", "1", "--warmup", "1",
            "--no-cpu-baseline", "--no-traffic"]
    one = run([sys.executable, "bench.py", "--gpus", "1"] + full)
    eight = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                 "--master-port", "29643", "bench.py", "--gpus", "8", "--backend", "gloo", "--same-device"] + full)
    assert eight["n_gpus"] == 8 and eight["config"]["starts_per_gpu"] == 8192 and eight["config"]["candidate_chunk"] == 8192
    assert "test_mode" in eight["config"] and eight["config"]["exchange"].startswith("torch.distributed gloo")
    pr = eight["per_rank"]
    assert [e["rank"] for e in pr] == list(range(8)) and all(e["starts"] == 8192 for e in pr)
    for e in pr:
        for k in ("local_ms_per_step", "exchange_wait_us_per_step", "exchange_us", "acq_gemm_frac", "acq_gemm_ms_per_step",
                  "evals_issued_per_step", "fit_ms_per_step"):
            assert k in e and e[k] >= 0, (k, e)
        assert 0 < e["acq_gemm_frac"] < 1 and 0 < e["evals_issued_per_step"] <= 8192 * 3
    sk = eight["skew"]
    assert sk["max_minus_min_ms"] == pytest.approx(sk["local_ms_max"] - sk["local_ms_min"]) and 0 <= sk["slowest_rank"] < 8
    assert abs(sum(e["evals_issued_per_step"] for e in pr) - eight["config"]["evals_issued_per_step"]) < 1e-6
    assert eight["config"]["evals_issued_per_step"] == one["config"]["evals_issued_per_step"]      # a start's path does not depend on its shard
    assert eight["result"]["best_index"] == one["result"]["best_index"]
    if all(e["potrf_fallbacks"] == 0 for e in pr):
        assert eight["result"]["best_value"] == one["result"]["best_value"]
        np.testing.assert_array_equal(eight["result"]["best_x"], one["result"]["best_x"])
    else:
        # eight processes on ONE GPU: a rank whose single-launch Cholesky could not have all its workgroups resident repeated the fit
        # on the multi-launch schedule, whose update chunks differ (nbo 4 against 3): the factor agrees to rounding, not in every bit.
        # One rank per GPU (the real run) never takes that path.
        assert eight["result"]["best_value"] == pytest.approx(one["result"]["best_value"], rel=1e-11)
        np.testing.assert_allclose(eight["result"]["best_x"], one["result"]["best_x"], rtol=0, atol=1e-9)


def test_nccl_on_distinct_devices_must_not_fall_back_silently():
    """bench.py: with --backend nccl and one rank per device the exchange is RCCL or the run fails; only the same-device test mode
    (and an explicit SLS_BENCH_ALLOW_GLOO_FALLBACK=1) may label a gloo exchange (CPU: the source says so)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "refusing to fall back to gloo" in src and "SLS_BENCH_ALLOW_GLOO_FALLBACK" in src
    assert 'assert exchange.startswith("ncclAllGather inside libsls_hip")' in src


@pytest.mark.gpu
def test_bench_line_carries_measured_traffic():
    """roofline.traffic is measured by the run itself (two rocprofv3 --pmc passes over a child of the same script and library
    behind the timed region), not quoted from a file: a number, with the launch shape and the correction it belongs to."""
    if not (os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 not installed")
    small = [a for a in SMALL if a != "--no-traffic"]
    j = run([sys.executable, "bench.py", "--gpus", "1"] + small)
    r = j["roofline"]
    assert isinstance(r["traffic"], float) and r["traffic"] > 0, r["traffic_detail"]
    d = r["traffic_detail"]
    assert d["launches_measured"] >= 1 and d["candidates_per_launch"] == 3072 and d["traffic_over_algorithmic"] > 0.5, d


def test_bench_self_spawn_arguments_round_trip(monkeypatch):
    """The argument list bench.py hands to the ranks it spawns parses back to the same namespace (CPU)."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--n", "300", "--d", "5", "--starts", "77", "--kernel", "se",
                                      "--same-device", "--no-cpu-baseline", "--steps", "3"])
    a = bench.parse()
    monkeypatch.setattr(sys, "argv", ["bench.py"] + bench.canonical_argv(a))
    assert vars(bench.parse()) == vars(a)


@pytest.mark.gpu
def test_bench_survives_an_unusable_rccl_communicator():
    """bench.py --gpus N creates its RCCL communicator (inside libsls_hip) on a watchdog thread and tries it once before
    the timed region.  Two ranks forced onto GPU 0 is a configuration RCCL refuses (one rank per GPU) or never completes:
    every rank must then agree on the rendezvous group for the exchange, say so in config.exchange, and still produce the
    single-rank answer."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", SLS_BENCH_TRY_RCCL_SAME_DEVICE="1",
               SLS_BENCH_RCCL_TIMEOUT="30")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", "bench.py", "--gpus", "2", "--backend", "nccl", "--same-device"] + SMALL
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    two = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    one = run([sys.executable, "bench.py", "--gpus", "1"] + SMALL)
    ex = two["config"]["exchange"]
    assert ("RCCL communicator unavailable" in ex) or ("ncclAllGather inside libsls_hip" in ex), ex
    assert two["result"]["best_index"] == one["result"]["best_index"] and two["result"]["best_value"] == one["result"]["best_value"]
