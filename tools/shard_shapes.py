#!/usr/bin/env python3
"""Per-GPU shard shapes of an 8 / 4 / 2 / 1-GPU run of the C4 workload, each measured ON ONE GPU (bench.py --starts S): what one
rank computes per step.  NOT a scaling curve: RCCL bootstrap, rank skew (a step ends when the slowest rank's active set is
empty) and the exchange are not in these numbers -- no multi-GPU hardware has been available.  -> gpurun_out/shard_shapes.json"""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"note": "single-GPU runs of one rank's share of the 65 536-start workload; exchange and rank skew not measured (no multi-GPU hardware)",
       "exchange": "not measured", "shards": []}
full = None
for gpus, starts in ((1, 65536), (2, 32768), (4, 16384), (8, 8192)):
    p = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--starts", str(starts), "--no-cpu-baseline", "--no-traffic", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900)
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    st = j["stage_ms_per_step"]
    fit = sum(st[k] for k in ("gram", "potrf", "trtri", "lauum"))
    if gpus == 1:
        full = j["ms_per_step"]
    out["shards"].append({"n_gpus_this_is_a_rank_of": gpus, "starts_per_gpu": starts, "ms_per_step": j["ms_per_step"],
                          "acq_gemm_frac_of_fp64_mfma_peak": j["roofline"]["frac"], "fit_ms_replicated_on_every_rank": fit,
                          "other_stages_ms": {k: st[k] for k in ("cross_gram", "grad_gemm", "finalize", "lbfgs")},
                          "single_gpu_step_over_this_shard": full / j["ms_per_step"]})
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(R, "gpurun_out", "shard_shapes.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
