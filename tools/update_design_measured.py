#!/usr/bin/env python3
"""Rewrite the measured table of DESIGN.md section 8 from profiles/r01_bench_line_1gpu.json + r01_pmc_acq_gemm.json."""
import json
b = json.load(open("profiles/r01_bench_line_1gpu.json"))
p = json.load(open("profiles/r01_pmc_acq_gemm.json"))
import os
aw = json.load(open("profiles/r01_cpu_as_written.json")) if os.path.exists("profiles/r01_cpu_as_written.json") else None
aw_row = ""
if aw:
    aw_row = (f'| reference call structure *as written* on the CPU (`tools/time_as_written.py`; **extrapolated**) | one evaluation '
              f'(value + gradient, each recomputing `PredictMaximumPointFromData` = N GEMVs) measured single-threaded at N = '
              + ", ".join(f'{r["N"]}: {r["s_per_eval"]:.3g} s' for r in aw["measured"])
              + f'; fitted exponent {aw["fit"]["exponent"]:.2f}; extrapolated to N = 8192: {aw["N8192_s_per_eval_1thread"]:.0f} s per evaluation per thread, '
              f'i.e. {aw["N8192_evals_per_s_all_cores"]:.2g} evaluations/s on the {aw["cores"]} cores it was timed on - the C4 step is not reachable in that structure '
              f'(the hoisted oracle above is the algorithm-equal baseline) |\n')
cf = json.load(open("profiles/r01_configs.json"))
c1, c2, c3, c5 = (cf["C1_bayesian_optimization_1d_20_iterations"], cf["C2_gp_fit_predict_N2048_D16_M4096"],
                  cf["C3_sequential_line_search_nd_D32_30_iterations"], cf["C5_map_objective_gradient_N4096_D128"])
s = open("DESIGN.md").read()
start = s.index("| quantity (C4:")
end = s.index("History of the dominant kernel this round")
st, sr, cb = b["stage_ms_per_step"], b.get("stage_rooflines", {}), b.get("cpu_baseline", {})
new = f'''| quantity (C4: N=8192, D=64, Matern-5/2, 65 536 starts x 50 evals, 1 GPU) | value |
|---|---|
| step time (fit + maximisation) | {b["ms_per_step"]/1e3:.2f} s |
| candidate evaluations / s | {b["value"]:.3g}; CPU oracle port on {cb.get("cores", "?")} host threads at the same N: {cb.get("value", float("nan")):.3g} (implied CPU step {cb.get("implied_step_seconds", float("nan")):.0f} s) |
{aw_row}| `acq_gemm_kernel` per launch ({int(b["roofline"]["flops_per_launch"] / (2 * 8192.0 * 8192.0))} candidates, {b["roofline"]["flops_per_launch"]/1e12:.1f} TFLOP) | {b["roofline"]["avg_launch_ms"]:.1f} ms (HIP events) / {p["avg_duration_ms"]:.1f} ms (rocprofv3) = **{b["roofline"]["achieved"]:.1f} TFLOP/s = {b["roofline"]["frac"]:.3f} of the 78.6 TFLOP/s fp64 MFMA peak** |
| PMC: `SQ_VALU_MFMA_BUSY_CYCLES` / MFMA = {p["mfma_busy_cycles_per_instruction"]:.1f}; MFMA pipe busy {100*p["mfma_busy_fraction"]:.1f} % of `GRBM_GUI_ACTIVE`; effective clock {p["effective_clock_GHz"]:.2f} GHz | the kernel is MFMA-issue bound, not clock- or HBM-bound |
| fabric traffic per launch (`FETCH_SIZE` x 2 + `WRITE_SIZE`, `roofline.traffic`) | {p["hbm_bytes_per_launch"]/1e9:.0f} GB vs {p["algorithmic_bytes_per_launch"]/1e9:.1f} GB algorithmic ({p["hbm_bytes_per_launch"]/1e9*16384/p["candidates_per_launch"]:.0f} GB per 16 384 candidates). Each 128x128 tile streams its two 8 MB operand panels; the tiles resident on an XCD share them through its 4 MB L2, which only works while the co-resident sharers of a panel stay within ~16 slabs (4 MB / 256 KB per slab step) of each other in k. The kernel therefore runs persistently (512 workgroups) with per-XCD generation gates (bounded spin). Measured per 65 536-candidate launch (`tools/prof_stagger.sh`): ungated 124.9 ms / 333 GB (hit rate 0.42; 0.34-0.85 across builds of this round with an identical k loop, depending on how the tiles drift apart); one gate per XCD 124.8 ms / 77 GB (0.856; the 8x8 tile group shares 16 panels, ideal 7/8); two gate groups per XCD started 20 us apart, so that the two workgroups of a CU (slots `s`, `s+32`: `cu_probe`) never run their epilogues together: **123.2 ms / 112 GB (0.795; 8x4 tiles share 12 panels)** - the default. A 256-wide tile is the next lever. `FETCH_SIZE` counts Infinity-Cache hits too, so DRAM traffic is lower. |
| other stages per step (ms) | cross_gram {st["cross_gram"]:.0f} ({sr.get("cross_gram", {}).get("achieved_GBps", 0)/1e3:.1f} TB/s written, transcendental-bound), grad_gemm {st["grad_gemm"]:.0f} ({sr.get("grad_gemm", {}).get("achieved_GBps", 0)/1e3:.1f} TB/s read), lbfgs {st["lbfgs"]:.0f}, finalize {st["finalize"]:.0f}; fit: gram {st["gram"]:.2f}, potrf {st["potrf"]:.1f} ({sr.get("potrf", {}).get("achieved_TFLOPs", 0):.0f} TFLOP/s, latency-bound diagonal chain), trtri {st["trtri"]:.1f} ({sr.get("trtri", {}).get("achieved_TFLOPs", 0):.0f} TFLOP/s), lauum {st["lauum"]:.1f} ({sr.get("lauum", {}).get("achieved_TFLOPs", 0):.0f} TFLOP/s) |
| C2 (N=2048, D=16, SE): fit / 4096-point predict | {c2["fit_ms_wall_incl_upload"]:.1f} ms / {c2["predict_ms_wall_incl_pcie"]:.2f} ms wall incl. PCIe (oracle on the host: {c2["cpu_oracle_fit_s"]:.1f} s / {c2["cpu_oracle_predict_s"]:.2f} s) |
| C5 (N=4096, D=128, Matern): MAP objective + gradient | {c5["ms_per_evaluation"]:.1f} ms per evaluation (8.2 ms before the diagonal-block kernel rewrite and the 128x64 trtri tiles) |
| C3 (`sequential_line_search_nd 32 30`) | {c3["ms_per_submit_mean"]:.0f} ms per `SubmitFeedbackData` on average (first call included; ~11 ms steady state; 61 ms before the per-start wavefront kernel); C1 (`bayesian_optimization_1d 1 20`): {c1["wall_s"]:.2f} s for 20 iterations (0.93 s with the ~20-launch tiled MAP evaluation, before `nll_small_kernel`) |
| parity vs the oracle (`profiles/r01_parity_report.md`) | max relative error 1e-16 .. 3e-11 on mu, sigma, gradients, EI, UCB; 4e-11 on EI; chosen maximiser x within 1.1e-8 |

'''
s = s[:start] + new + s[end:]
open("DESIGN.md", "w").write(s)
print("updated")
