#!/usr/bin/env python3
"""Which starts of the multi-start maximiser end at a different value on the HIP path than on the oracle, and why.

For every start: the first evaluation budget n at which the two y_stars differ by more than 1e-6 relative, the oracle's
closest Armijo margin up to that evaluation (slso_acq_maximize_diag) and the distance between the two end points.
A start whose Armijo test sat at rounding level right before the first difference took the other branch of the same
algorithm because the two implementations sum in a different order -- not a disagreement of the objective.

usage: python tools/diverging_starts.py [D N S n_local]     (GPU box; writes gpurun_out/diverging_starts.json)"""
import importlib
import os as _os
_os.environ.setdefault("OMP_NUM_THREADS", "64")   # the oracle stops scaling past ~64 threads
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_py as orc  # noqa: E402
from util import synth_candidates, synth_problem  # noqa: E402

sls = importlib.import_module("sequential-line-search_amd")


def analyse(ctx, D, N, S, n_local, kernel, acq, wave):
    os.environ["SLS_WAVE_PATH"] = "1" if wave else "0"
    sls.tuning_reload()
    X, y, theta, b = synth_problem(orc, D, N)
    starts = synth_candidates(orc, D, S)
    ref = orc.Regressor(X, y, theta, b, kernel=kernel)
    gp = sls.GP(ctx, X, y, theta, b, kernel)
    first = np.full(S, -1)
    for n in range(1, n_local + 1):
        ro = ref.acq_maximize(starts, n, acq, 2.0)
        rg = gp.acq_maximize(starts, n, acq, 2.0)
        bad = ~np.isclose(rg["y_stars"], ro["y_stars"], rtol=1e-6, atol=1e-12)
        first[(first < 0) & bad] = n
    rd = ref.acq_maximize(starts, n_local, acq, 2.0, diag=True)
    rg = gp.acq_maximize(starts, n_local, acq, 2.0)
    final_bad = ~np.isclose(rg["y_stars"], rd["y_stars"], rtol=1e-6, atol=1e-12)
    rows = []
    for i in np.nonzero(first > 0)[0]:
        n1 = int(first[i])
        # margin of the oracle's Armijo tests up to the evaluation before the first difference
        rm = ref.acq_maximize(starts[:, i:i + 1], n1, acq, 2.0, diag=True)
        rows.append(dict(start=int(i), first_diff_eval=n1, armijo_margin_before=float(rm["armijo_margin"][0]),
                         armijo_eval=int(rm["armijo_eval"][0]), still_differs_at_end=bool(final_bad[i]),
                         y_gpu=float(rg["y_stars"][i]), y_oracle=float(rd["y_stars"][i]),
                         dx_end=float(np.abs(rg["x_stars"][:, i] - rd["x_stars"][:, i]).max())))
    gp.close()
    return dict(D=D, N=N, S=S, n_local=n_local, kernel=kernel, acq=acq, path="wave" if wave else "tiled",
                n_diverged_any_time=int((first > 0).sum()), n_differ_at_end=int(final_bad.sum()), starts=rows)


def main():
    a = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else [5, 120, 96, 25]
    ctx = sls.Context(0)
    out = []
    for kernel in (0, 1):
        for acq in (0, 1):
            for wave in (True, False):
                r = analyse(ctx, *a, kernel, acq, wave)
                out.append(r)
                ms = [s["armijo_margin_before"] for s in r["starts"]]
                print(f"kernel {kernel} acq {acq} {r['path']:5s}: {r['n_differ_at_end']:3d}/{a[2]} differ at the end, "
                      f"{r['n_diverged_any_time']} at some budget; Armijo margins before the first difference: "
                      f"max {max(ms) if ms else 0:.2e}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diverging_starts.json"), "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
