#!/usr/bin/env python3
"""<prefix>_pmc_by_kernel.json -> <prefix>_pmc_acq_gemm.json (the `traffic` source of bench.py).
usage: tools/make_pmc_summary.py [prefix = profiles/r01]"""
import json
import sys
PFX = sys.argv[1] if len(sys.argv) > 1 else "profiles/r01"
d = json.load(open(PFX + "_pmc_by_kernel.json"))
k = [x for x in d if x.startswith("acq_gemm_kernel")][0]
c = d[k]
fetch_kb, write_kb = c["FETCH_SIZE"]["mean"], c["WRITE_SIZE"]["mean"]
dur = c["duration_ns[pmc_mfma]"]["mean"]
gui = c["GRBM_GUI_ACTIVE"]["mean"] / 8                      # summed over the 8 XCDs
mfma_busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024    # summed over 256 CU x 4 SIMD
flops = c["SQ_INSTS_VALU_MFMA_MOPS_F64"]["mean"] * 512
N = 8192
cand = int(round(flops / (2.0 * N * N)))                     # candidates per launch (bench.py --chunk)
out = {
    "kernel": k, "launches_profiled": c["FETCH_SIZE"]["n"],
    "command": "rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 1 --warmup 1 --n-local 6 --no-cpu-baseline  (tools/prof_r0N.sh; separate passes for FETCH_SIZE, WRITE_SIZE, SQ/GRBM; round 2: SLS_COMPACT=0 so that every dispatch has the full 65 536-candidate shape)",
    "avg_duration_ms": dur / 1e6,
    "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
    "fetch_bytes_corrected": fetch_kb * 1024 * 2, "write_bytes": write_kb * 1024,
    "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
    "correction_note": "MI355X_MICROARCH.md 'HBM': on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced read -> doubled; WRITE_SIZE used as is (matches the P tile + partials written per launch). FETCH_SIZE counts L2->fabric requests incl. Infinity-Cache hits, so this is an upper bound on DRAM traffic.",
    "candidates_per_launch": cand,
    "algorithmic_bytes_per_launch": 2 * cand * N * 8 + N * N * 8 + cand * N * 8,
    "mfma_flops_counted": flops, "effective_clock_GHz": gui / dur,
    "mfma_busy_fraction": mfma_busy / gui, "mfma_busy_cycles_per_instruction": c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (flops / 2048),
}
json.dump(out, open(PFX + "_pmc_acq_gemm.json", "w"), indent=1)
print({k2: out[k2] for k2 in ("avg_duration_ms", "FETCH_SIZE_KB_raw", "hbm_bytes_per_launch", "mfma_busy_fraction", "effective_clock_GHz")})
