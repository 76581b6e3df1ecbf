#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: GP-fit + acquisition-maximisation step at N=8192, D=64 (config C4).

One STEP = (a) GP fit on the device-resident design matrix: Gram (Matern-5/2) + Cholesky + K^-1 + alpha + mu+,
           (b) multi-start EI maximisation: 65 536 random starts, each a bounded L-BFGS capped at 50 objective evaluations
               (value + gradient), advanced in lock step over the active set and sharded over the ranks,
           (c) one all-gather of (value, global index, x[D]) per rank and the first-maximum merge.
The cap is the reference's semantics (NLopt max_evals, src/acquisition-function.cpp:128-129): a start that can no longer
move stops consuming evaluations.  value = candidate evaluations ACTUALLY PERFORMED per second by the whole job (all
ranks; `config.evals_issued_per_step`, at most 65 536 x 50 = `config.evals_cap_per_step`); ms_per_step is the step time.
Strong scaling: the 65 536 starts are fixed and split over the ranks; every rank repeats the (cheap) fit.

Launch: python bench.py [--gpus N --steps K --warmup W].  For N > 1 the ranks are one process per GPU: either the caller
starts them (torch.distributed.run --nproc-per-node N ... bench.py --gpus N, the driver's form), or -- invoked plainly --
bench.py starts that same torch.distributed.run itself on 127.0.0.1 and a free port and relays rank 0's JSON line
(`config.launcher` says which).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL between processes needs dmabuf IPC on this host driver (hipIpcGetMemHandle fails otherwise); read by the HSA runtime when
# it initialises, i.e. at the first HIP call below -- a launcher that did not export it still gets it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X dense fp64 matrix peak: 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz
                               # (v_mfma_f64_16x16x4_f64 = 2048 FLOP / 64 clk / SIMD; MI355X_MICROARCH.md gives no
                               #  fp64 row -- derived from the f32 row's 64 FLOP/clk/SIMD halved)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--n", "--num-train", dest="n", type=int, default=8192, help="training points N")
    p.add_argument("--d", "--dims", dest="d", type=int, default=64, help="dimensions D")
    p.add_argument("--starts", type=int, default=65536, help="total multi-start count S (split over ranks)")
    p.add_argument("--n-local", type=int, default=50, help="objective evaluations per start")
    p.add_argument("--kernel", choices=["matern52", "se"], default="matern52")
    p.add_argument("--chunk", type=int, default=65536,
                   help="candidates per device pass (K*, C*, P workspaces: 3 x chunk x N x 8 B = 12.9 GB at the default)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                   help="collective backend; gloo + --same-device exercises the N>1 path on a 1-GPU box (tests only)")
    p.add_argument("--same-device", action="store_true", help="all ranks use GPU 0 (tests only; never a bench line)")
    p.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes behind roofline.traffic")
    p.add_argument("--traffic-child", action="store_true",
                   help="internal: what measure_traffic() runs under rocprofv3 (one fit + a few full-shape rounds of the maximiser)")
    return p.parse_args()


def synth(D, N, S):
    """SURVEY.md 8(d) synthetic inputs (seeds 1234/1235/1236)."""
    X = np.random.default_rng(1234).uniform(0.0, 1.0, (D, N))
    y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * np.random.default_rng(1235).normal(size=N)
    theta = np.concatenate([[0.5], np.full(D, 0.5 * np.sqrt(max(D, 8) / 8.0))])
    starts = np.random.default_rng(1236).uniform(0.0, 1.0, (D, S))
    return np.asfortranarray(X), y, theta, 0.005, np.asfortranarray(starts)


def cpu_baseline(args, kernel_id):
    """The oracle ("port", hoisted mode) timed on the host cores on a bounded sample of the SAME workload: the full-size
    fit (N training points) and 1024 of the starts x 2 lock-step evaluations (about 35 s of CPU work at N = 8192)."""
    from oracle import oracle_py as orc
    orc.build()
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    Ss, evals = min(1024, args.starts), 2
    X, y, theta, b, starts = synth(args.d, args.n, Ss)
    t0 = time.perf_counter()
    ref = orc.Regressor(X, y, theta, b, kernel=kernel_id)
    t_fit = time.perf_counter() - t0
    t0 = time.perf_counter()
    ro = ref.acq_maximize(starts, evals, n_threads=threads)
    t_acq = time.perf_counter() - t0
    rate = Ss * evals / t_acq
    step_s = t_fit + args.starts * args.n_local / rate
    mu_o, sg_o = ref.predict_batch(starts[:, :256])
    oracle_out = dict(X=X, y=y, theta=theta, b=b, starts=starts, evals=evals, y_stars=ro["y_stars"], x_stars=ro["x_stars"],
                      value=ro["value"], x=ro["x"], mu=mu_o, sigma=sg_o)
    return oracle_out, {
        "value": rate, "unit": "candidate-evals/s", "cores": threads, "kind": "port",
        "sample": (f"oracle (hoisted mode: Cholesky, cached alpha and mu+, blocked K^-1 k) at the full N={args.n}, D={args.d}: "
                   f"fit {t_fit:.1f} s; {Ss} of the {args.starts} starts x {evals} evaluations in {t_acq:.2f} s; "
                   f"{threads} OpenMP threads of a {cores}-core host; implied CPU step (fit + {args.starts} x "
                   f"{args.n_local} evals) = {step_s:.0f} s"),
        "fit_seconds": t_fit, "implied_step_seconds": step_s,
    }


def cpu_as_written(args, kernel_id, cores):
    """BASELINE.md 5, mode 1: the reference's AS-WRITTEN call structure -- one NLopt callback = derivative then value
    (src/acquisition-function.cpp:38-56), each recomputing PredictMaximumPointFromData = N x PredictMu with K^-1 y not cached
    (src/regressor.cpp:29-43, src/acquisition-function.cpp:188): O(N^3) per evaluation.  Feasible only at small N: timed single-
    threaded through the oracle's as-written entry points at N in {256, 512, 1024} (D as the workload's), a power law fitted and
    EXTRAPOLATED to the workload's N; the reference's multi-start loop runs one evaluation per hardware thread, so the whole-machine
    rate is cores / t.  Labelled extrapolated: it is context beside the measured hoisted port, never a measured number."""
    from oracle import oracle_py as orc
    os.environ["OMP_NUM_THREADS"] = "1"
    rows = []
    for N, reps in ((256, 3), (512, 2), (1024, 1)):
        rng = np.random.default_rng(1234 + N)
        X = np.asfortranarray(rng.uniform(0, 1, (args.d, N)))
        y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * rng.standard_normal(N)
        theta = np.concatenate([[0.5], np.full(args.d, 0.5 * np.sqrt(max(args.d, 8) / 8.0))])
        ref = orc.Regressor(X, y, theta, 0.005, kernel=kernel_id)
        x = rng.uniform(0, 1, args.d)
        t0 = time.perf_counter()
        for _ in range(reps):
            ref.acq_derivative_as_written(x)
            ref.acq_value_as_written(x)
        rows.append({"N": N, "s_per_eval_1thread": (time.perf_counter() - t0) / reps, "reps": reps})
    p, logc = np.polyfit(np.log([r["N"] for r in rows]), np.log([r["s_per_eval_1thread"] for r in rows]), 1)
    t_n = float(np.exp(logc) * args.n ** p)
    return {"kind": "port of the as-written call structure", "extrapolated": True, "measured": rows, "fit_exponent": float(p),
            "N": args.n, "s_per_eval_1thread": t_n, "value": cores / t_n, "unit": "candidate-evals/s", "cores": cores,
            "implied_step_seconds": args.starts * args.n_local * t_n / cores,
            "note": "one evaluation per hardware thread (parallel-util's queue); power law through N = 256, 512, 1024 evaluated at the workload's N"}


def parity_vs_oracle(sls, ctx, kernel_id, o):
    """The oracle numbers of the cpu_baseline leg (full N, 1024 starts x 2 evaluations, mu / sigma at 256 points) against
    the HIP path on the same inputs: maximum relative errors (north_star bar: 1e-6)."""
    gp = sls.GP(ctx, o["X"], o["y"], o["theta"], o["b"], kernel_id)
    rg = gp.acq_maximize(o["starts"], o["evals"])
    mu, sg = gp.predict(o["starts"][:, :256])
    gp.close()

    def rel(a, b, floor):
        a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
        return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
    ys = np.abs(o["y_stars"]).max()
    out = {"mu": rel(mu, o["mu"], 1e-3 * np.abs(o["mu"]).max()), "sigma": rel(sg, o["sigma"], 1e-3 * np.abs(o["sigma"]).max()),
           "y_stars": rel(rg["y_stars"], o["y_stars"], 1e-6 * ys), "x_stars_abs": float(np.abs(rg["x_stars"] - o["x_stars"]).max()),
           "best_value": rel(rg["value"], o["value"], 1e-300), "best_x_abs": float(np.abs(rg["x"] - o["x"]).max()),
           "n_points": 256, "n_starts": int(o["starts"].shape[1]), "evals_per_start": int(o["evals"])}
    return out, max(out["mu"], out["sigma"], out["y_stars"], out["best_value"])


def traffic_child(args):
    """One fit + a 4-evaluation maximiser run of the SAME library, shapes and launch parameters as the timed region, with
    SLS_COMPACT=0 (set by the parent): every acq_gemm launch has the full candidate shape.  Runs under rocprofv3 --pmc."""
    sls = importlib.import_module("sequential-line-search_amd")          # no torch here: the library owns the device
    kernel_id = sls.KERNEL_MATERN52 if args.kernel == "matern52" else sls.KERNEL_SE
    X, y, theta, b, starts = synth(args.d, args.n, args.starts)
    ctx = sls.Context(0)
    ctx.set_candidate_chunk(args.chunk)
    gp = sls.GP(ctx, X, y, theta, b, kernel_id)
    gp.acq_maximize(starts, 4)
    gp.close()
    ctx.close()


def measure_traffic(args):
    """roofline.traffic: memory-side bytes of ONE full-shape acq_gemm launch, from two rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots") over a child run of this same script and library.
    Corrections as that guide prescribes: FETCH_SIZE (KB) x 2 on gfx950 for wide coalesced reads (the kernel's loads are
    global_load_lds_dwordx4, 16 B per lane), WRITE_SIZE (KB) as reported.  The counters sit on the L2's fabric side: Infinity-Cache
    hits are included, so the figure is an upper bound on DRAM traffic.  Returns (bytes or None, detail dict)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, {"skipped": "rocprofv3 not found"}
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or os.environ.get("HSA_TOOLS_LIB"):
        return None, {"skipped": "this run is itself under a profiler: no nested counter collection"}
    raw, dur, n_launch = {}, [], 0
    child = [sys.executable, os.path.abspath(__file__), "--traffic-child", "--num-train", str(args.n), "--dims", str(args.d),
             "--starts", str(args.starts), "--kernel", args.kernel, "--chunk", str(args.chunk)]
    env = dict(os.environ, SLS_COMPACT="0", TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="sls_pmc_", dir="/tmp")
        try:
            p = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, {"skipped": f"rocprofv3 --pmc {counter} failed (rc {p.returncode}): {(p.stderr or p.stdout)[-300:]}"}
            per = {}
            for r in csv.DictReader(open(files[0])):
                if "acq_gemm" in r["Kernel_Name"] and r["Counter_Name"] == counter:   # acq_gemm_kernel, or acq_gemm_half_kernel for a small launch
                    k = r["Dispatch_Id"]
                    e = per.setdefault(k, [0.0, int(r["Grid_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
                    e[0] += float(r["Counter_Value"])
            if not per:
                return None, {"skipped": f"no acq_gemm dispatch in the {counter} pass"}
            gmax = max(v[1] for v in per.values())
            full = [v for v in per.values() if v[1] == gmax]
            raw[counter] = sum(v[0] for v in full) / len(full)
            dur += [v[2] for v in full]
            n_launch = len(full)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    N = args.n
    cand = min(args.chunk, (args.starts + 127) // 128 * 128)
    fetch_b, write_b = raw["FETCH_SIZE"] * 1024.0 * 2.0, raw["WRITE_SIZE"] * 1024.0
    alg = 2.0 * cand * N * 8 + N * N * 8.0 + cand * N * 8.0        # read K*, C*; read K^-1 once; write P
    detail = {"candidates_per_launch": cand, "launches_measured": n_launch, "avg_launch_ms_under_profiler": sum(dur) / len(dur) / 1e6,
              "FETCH_SIZE_KB_raw": raw["FETCH_SIZE"], "WRITE_SIZE_KB_raw": raw["WRITE_SIZE"], "fetch_bytes_corrected": fetch_b,
              "write_bytes": write_b, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fetch_b + write_b) / alg,
              "correction": "FETCH_SIZE x 2 (gfx950, 16 B/lane coalesced reads), WRITE_SIZE as reported; fabric-side counters: "
                            "Infinity-Cache hits included (upper bound on DRAM traffic)",
              "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-trace -- python bench.py --traffic-child ... (SLS_COMPACT=0: "
                         "every launch has the full candidate shape), run by this bench.py after its timed region"}
    return fetch_b + write_b, detail


def baseline_metric():
    """BASELINE.json's metric string, verbatim (value = its candidate-evals/sec part, ms_per_step = its step-time part)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "GP-fit+acq-max step time (ms) and candidate-evals/sec at N=8192 D=64"


def stage_rooflines(prof, N, D, Np, cand, matern):
    """Secondary kernels against their own bound (algorithmic bytes / flops of DESIGN.md 4 per launch group)."""
    def per_launch_ms(name):
        ms, n = prof[name]
        return ms / n if n else float("nan")
    out = {}
    t = per_launch_ms("cross_gram")          # writes K* (and C*): 8 N S_c bytes each; HBM-write bound
    out["cross_gram"] = {"bound": "hbm", "achieved_GBps": (2 if matern else 1) * 8.0 * N * cand / (t * 1e-3) / 1e9, "peak_GBps": 8000.0}
    t = per_launch_ms("grad_gemm")           # reads P and C*: 16 N S_c bytes; 4 N D S_c flops, i.e. D / 4 flop per byte: the
    # binding roofline is the one with the LONGER minimum time (fp64 MFMA from D >= 40: ridge 78.6 TFLOP/s / 8 TB/s = 9.8 flop/B)
    gg_bytes, gg_flops = 16.0 * N * cand, 4.0 * N * D * cand
    out["grad_gemm"] = {"bound": "mfma" if gg_flops / (PEAK_FP64_MFMA_TFLOPS * 1e12) >= gg_bytes / 8e12 else "hbm",
                        "achieved_GBps": gg_bytes / (t * 1e-3) / 1e9, "peak_GBps": 8000.0,
                        "achieved_TFLOPs": gg_flops / (t * 1e-3) / 1e12, "peak_TFLOPs": PEAK_FP64_MFMA_TFLOPS}
    t = per_launch_ms("gram")                # writes the lower triangle of K_y: 4 N^2 bytes
    out["gram"] = {"bound": "hbm", "achieved_GBps": 4.0 * N * N / (t * 1e-3) / 1e9, "peak_GBps": 8000.0}
    # potrf N^3/3, triangular inverse N^3/3 (the recursive doubling executes ~N^3/3 MFMA flops: its GEMMs run over
    # triangular k ranges), lauum N^3/3: the three stages of K^-1 = (L L^T)^-1 sum to the N^3 of SURVEY.md 8(d)
    t_fit = 0.0
    fused = prof.get("potri", (0.0, 0))[1] > 0          # N <= 4096: factor + inverse in ONE launch (kernels_chol.hip: potri_team)
    stages = (("potri", float(N) ** 3),) if fused else (("potrf", N ** 3 / 3.0), ("trtri", N ** 3 / 3.0), ("lauum", N ** 3 / 3.0))
    for name, flops in stages:
        t = per_launch_ms(name)
        t_fit += t
        out[name] = {"bound": "mfma", "achieved_TFLOPs": flops / (t * 1e-3) / 1e12, "peak_TFLOPs": PEAK_FP64_MFMA_TFLOPS}
    # north_star's ">= 50 % on the Gram + Cholesky + predict pipeline" target is stated on the fit chain as a whole
    t_fit += per_launch_ms("gram")
    out["fit_pipeline"] = {"bound": "mfma", "stages": "gram + potri (fused)" if fused else "gram + potrf + trtri + lauum", "ms": t_fit,
                           "achieved_TFLOPs": (N ** 3 + N * N * D) / (t_fit * 1e-3) / 1e12, "peak_TFLOPs": PEAK_FP64_MFMA_TFLOPS}
    for v in out.values():
        v["frac"] = v["achieved_TFLOPs"] / v["peak_TFLOPs"] if v["bound"] == "mfma" else v["achieved_GBps"] / v["peak_GBps"]
    return out


def canonical_argv(args):
    """The parsed arguments under their long names (torch.distributed.run's own parser prefix-matches a script argument
    such as `--n` against its options even behind the script path)."""
    out = ["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--num-train", str(args.n),
           "--dims", str(args.d), "--starts", str(args.starts), "--n-local", str(args.n_local), "--kernel", args.kernel,
           "--chunk", str(args.chunk), "--backend", args.backend]
    return (out + (["--no-cpu-baseline"] if args.no_cpu_baseline else []) + (["--same-device"] if args.same_device else []) +
            (["--no-traffic"] if args.no_traffic else []))


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks the way the driver would (one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 and a free port), pass every argument through, relay the output."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SLS_BENCH_LAUNCHER="self")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + canonical_argv(args)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.traffic_child:
        return traffic_child(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            raise SystemExit(self_spawn(args))
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = None                                          # the rendezvous group exchanges host tensors
    if world > 1:
        # torch.distributed is the RENDEZVOUS only (gloo: barrier, the 128-byte RCCL id, the max-over-ranks clock); the
        # data-path collective of a step is the ncclAllGather inside libsls_hip (sls_comm_allgather_best)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    sls = importlib.import_module("sequential-line-search_amd")
    kernel_id = sls.KERNEL_MATERN52 if args.kernel == "matern52" else sls.KERNEL_SE
    D, N, S = args.d, args.n, args.starts
    lo, hi = sls.shard_range(S, rank, world)
    S_loc = hi - lo

    X, y, theta, b, starts = synth(D, N, S)
    ctx = sls.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_candidate_chunk(args.chunk)
    # inputs resident in HBM before the timed region
    X_dev = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)               # (N, D) C-order == D x N column-major
    y_dev = torch.from_numpy(y).to(dev)
    starts_dev = torch.from_numpy(np.ascontiguousarray(starts[:, lo:lo + S_loc].T)).to(dev)
    gp = sls.GP(ctx, X, y, theta, b, kernel_id)

    comm, comm_ctx, exchange = None, None, "none (1 GPU)"
    if world > 1:
        try_rccl = args.backend == "nccl" and (not args.same_device or os.environ.get("SLS_BENCH_TRY_RCCL_SAME_DEVICE") == "1")
        if try_rccl:
            # The communicator lives on its OWN context and is created (and tried once) on a watchdog thread: if RCCL's
            # bootstrap cannot complete on this node every rank learns so (all_reduce below) and the run fails with the reason,
            # instead of hanging in a collective; only the same-device test mode or an explicit override continues over gloo.
            import threading
            try:
                uid = [sls.Comm.unique_id() if rank == 0 else None]
            except sls.SlsError as e:
                uid = [None]
                print(f"[bench] rank 0: {e}", file=sys.stderr)
            dist.broadcast_object_list(uid, src=0)
            box = {}

            def make():
                try:
                    cctx = sls.Context(local_rank)
                    c = sls.Comm(cctx, uid[0], rank, world)
                    v, i, x = c.allgather_best(float(rank), rank, np.full(D, float(rank)))   # highest rank wins
                    assert (v, i) == (float(world - 1), world - 1) and x[0] == world - 1, (v, i)
                    box["comm"], box["ctx"] = c, cctx
                except Exception as e:                                                         # noqa: BLE001
                    box["err"] = repr(e)
            if uid[0] is not None:
                t = threading.Thread(target=make, daemon=True)
                t.start()
                t.join(timeout=float(os.environ.get("SLS_BENCH_RCCL_TIMEOUT", "90")))
                if t.is_alive():
                    box["err"] = "ncclCommInitRank / first ncclAllGather did not return within the watchdog timeout"
            else:
                box["err"] = "RCCL could not be loaded on rank 0"
            ok = torch.tensor([1.0 if "comm" in box and "err" not in box else 0.0], dtype=torch.float64)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() < 1.0:
                if not args.same_device and os.environ.get("SLS_BENCH_ALLOW_GLOO_FALLBACK") != "1":
                    # one rank per GPU over RCCL is what a multi-GPU bench line measures: without the communicator the run FAILS
                    # (a line whose exchange silently went through gloo would be read as an xGMI number)
                    raise SystemExit(f"[bench] rank {rank}: --backend nccl on distinct devices, but the RCCL communicator is unavailable "
                                     f"({box.get('err', 'see other ranks')}); refusing to fall back to gloo "
                                     "(SLS_BENCH_ALLOW_GLOO_FALLBACK=1 overrides, and labels the line)")
                exchange = ("torch.distributed gloo all_gather (RCCL communicator unavailable on some rank: "
                            f"{box.get('err', 'see other ranks')})")
            else:
                comm, comm_ctx = box["comm"], box["ctx"]
                exchange = "ncclAllGather inside libsls_hip (sls_comm_allgather_best), RCCL over xGMI"
        else:
            exchange = "torch.distributed gloo all_gather (test mode)"

    def exchange_once(r):
        if comm is not None:
            return comm.allgather_best(r["value"], r["index"], r["x"])
        return sls.exchange_best(r["value"], r["index"], r["x"], device=xdev)

    def step():
        t_a = time.perf_counter()
        gp.refit_dev(X_dev.data_ptr(), y_dev.data_ptr())
        r = gp.acq_maximize_dev(starts_dev.data_ptr(), S_loc, args.n_local, sls.ACQ_EI, 1.0, offset=lo)   # returns the winner: synchronous
        issued[0] += gp.last_stats()["evals_issued"]
        t_b = time.perf_counter()
        t_local[0] += t_b - t_a
        if world > 1:                                   # the single exchange of the step
            v, i, x = exchange_once(r)
            t_xchg[0] += time.perf_counter() - t_b      # includes waiting for the slowest rank
            return dict(value=v, index=i, x=x)
        return r

    issued, t_local, t_xchg = [0], [0.0], [0.0]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.prof_enable(True)
    ctx.prof_reset()
    issued[0], t_local[0], t_xchg[0] = 0, 0.0, 0.0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=xdev if xdev is not None else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ti = torch.tensor([float(issued[0])], dtype=torch.float64, device=xdev if xdev is not None else "cpu")
        dist.all_reduce(ti, op=dist.ReduceOp.SUM)
        issued_total = int(ti.item())
    else:
        issued_total = issued[0]
    names = ["gram", "potri", "potrf", "trtri", "lauum", "cross_gram", "acq_gemm", "grad_gemm", "finalize", "lbfgs"]
    prof = {n: ctx.prof_get(n) for n in names}
    ctx.prof_enable(False)
    per_rank = None
    if world > 1:
        # What makes a scaling curve diagnosable from ONE line: every rank's own numbers, all-gathered over the rendezvous group
        # (behind the timed region).  exchange_us: the collective alone, ranks aligned by a barrier in front of every repetition.
        reps = 20
        t_pure = 0.0
        for _ in range(reps):
            dist.barrier()
            t_x = time.perf_counter()
            exchange_once(res if "value" in res else dict(value=0.0, index=0, x=np.zeros(D)))
            t_pure += time.perf_counter() - t_x
        g_ms, g_n = prof["acq_gemm"]
        mine = {"rank": rank, "device": local_rank, "starts": S_loc, "local_ms_per_step": t_local[0] / args.steps * 1e3,
                "exchange_wait_us_per_step": t_xchg[0] / args.steps * 1e6, "exchange_us": t_pure / reps * 1e6,
                "evals_issued_per_step": issued[0] / args.steps,
                "acq_gemm_ms_per_step": g_ms / args.steps, "acq_gemm_launches_per_step": g_n / args.steps,
                "acq_gemm_frac": (2.0 * N * N * issued[0] / (g_ms * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS) if g_ms > 0 else 0.0,
                "fit_ms_per_step": sum(prof[n][0] for n in ("gram", "potri", "potrf", "trtri", "lauum")) / args.steps,
                "potrf_fallbacks": int(ctx.prof_get("potrf_fallbacks")[1])}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = sorted(gathered, key=lambda e: e["rank"])

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        evals_cap = S * args.n_local
        evals_issued = issued_total / args.steps            # whole job, per step
        Np = (N + 127) // 128 * 128
        gemm_ms, gemm_launches = prof["acq_gemm"]
        chunk = min(args.chunk, (S_loc + 127) // 128 * 128)
        # Dominant kernel: w = K^-1 k = 2 N^2 algorithmic flops per candidate evaluation (SURVEY.md 8(d) "EI value+grad, one
        # candidate-eval": 2 N^2 of the 2 N^2 + 6 N D).  Launches shrink with the active set, so the rate is taken over all
        # of rank 0's launches of the timed region: 2 N^2 x (evaluations rank 0 issued) / (its total acq_gemm time).
        issued_rank0 = issued[0]
        flops_total = 2.0 * N * N * issued_rank0
        achieved = flops_total / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        avg_ms = gemm_ms / max(gemm_launches, 1)
        # roofline.traffic: measured by THIS run, behind the timed region: two rocprofv3 --pmc passes over a child that runs the same
        # library at the same shapes (measure_traffic); bytes per full-shape launch (achieved is per launch of the timed region,
        # whose launches shrink with the active set: traffic_detail.candidates_per_launch says what the figure belongs to)
        traffic, traffic_detail = (None, {"skipped": "--no-traffic / multi-GPU run"}) if (args.no_traffic or world > 1) else measure_traffic(args)
        out = {
            "metric": baseline_metric(),
            "value": evals_issued / (ms_per_step * 1e-3), "unit": "candidate-evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4: multi-start EI maximisation", "N": N, "D": D, "starts_total": S,
                       "starts_per_gpu": S_loc, "n_local_evals": args.n_local, "kernel": args.kernel,
                       "candidate_chunk": chunk, "parallelism": f"starts sharded over {world} GPU(s), one all-gather", "exchange": exchange,
                       "launcher": ("single process" if world == 1 else "torch.distributed.run started by bench.py itself"
                                    if os.environ.get("SLS_BENCH_LAUNCHER") == "self" else "torch.distributed.run (caller)"),
                       "evals_cap_per_step": evals_cap, "evals_issued_per_step": evals_issued,
                       "evals_semantics": "n_local is a cap per start (NLopt max_evals); finished starts leave the batch",
                       "acq_gemm_form": ("2 workgroups per CU, persistent, generation-gated" if os.environ.get("SLS_ACQ_WG_PER_CU") == "2"
                                         else "1 workgroup per CU, persistent, gated every 16th generation")
                                        + " (defaults; SLS_ACQ_WG_PER_CU / SLS_PERSIST / SLS_GATE_EVERY)"},
            "roofline": {"bound": "mfma", "kernel": "acq_gemm_kernel", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_unit": "bytes per full-shape launch", "traffic_detail": traffic_detail, "avg_launch_ms": avg_ms, "launches": gemm_launches,
                         "flops_total": flops_total, "candidates_per_launch_avg": issued_rank0 / max(gemm_launches, 1)},
            "stage_ms_per_step": {n: prof[n][0] / args.steps for n in names},
            "stage_rooflines": stage_rooflines(prof, N, D, Np, issued_rank0 / max(prof["cross_gram"][1], 1), args.kernel == "matern52"),
            "result": {"best_value": res["value"], "best_index": int(res["index"]), "best_x": [float(v) for v in res["x"]]},
            # single-launch Cholesky factorisations that gave up (their workgroups could not all be resident) and were repeated on
            # the multi-launch schedule: 0 on a GPU this process has to itself
            "potrf_fallbacks": int(ctx.prof_get("potrf_fallbacks")[1]),
        }
        if per_rank is not None:
            loc = [e["local_ms_per_step"] for e in per_rank]
            out["per_rank"] = per_rank
            out["skew"] = {"local_ms_max": max(loc), "local_ms_min": min(loc), "max_minus_min_ms": max(loc) - min(loc),
                           "slowest_rank": int(np.argmax(loc)), "exchange_us_max": max(e["exchange_us"] for e in per_rank),
                           "note": "local = refit + this rank's shard of the maximisation (host clock around the synchronous calls); "
                                   "exchange_wait includes waiting for the slowest rank, exchange_us is the collective alone"}
            if args.backend == "nccl" and not args.same_device and os.environ.get("SLS_BENCH_ALLOW_GLOO_FALLBACK") != "1":
                assert exchange.startswith("ncclAllGather inside libsls_hip"), exchange
        if args.same_device or args.backend != "nccl":
            out["config"]["test_mode"] = "ranks share GPU 0 over gloo: not a bench line"
        if world == 1 and not args.no_cpu_baseline:
            oracle_out, out["cpu_baseline"] = cpu_baseline(args, kernel_id)
            out["cpu_baseline"]["as_written"] = cpu_as_written(args, kernel_id, os.cpu_count() or 1)
            # the oracle run is not thrown away: the same inputs go through the HIP path and the two are diffed
            out["parity"], out["parity_max_rel"] = parity_vs_oracle(sls, ctx, kernel_id, oracle_out)
        print(json.dumps(out))
    if comm is not None:
        comm.close()
        comm_ctx.close()
    gp.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
