"""The host C++ layer (reference public surface over the C ABI) on the GPU: behavioural checks + the two CLI scenarios."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "sequential-line-search_amd", "bin")


def run(name, *args, timeout=600):
    p = subprocess.run([os.path.join(BIN, name), *map(str, args)], capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout


def test_host_classes():
    out = run("test_host")
    assert "HOST TESTS PASSED" in out, out[-3000:]


def test_bayesian_optimization_1d_demo():
    """BASELINE config C1: 20 iterations; true optimum x = 0.852733, f = 2.273928.  With the reference's
    default maximiser branch (DIRECT -> L-BFGS, host/direct.cpp) and DIRECT(300) as the global phase of the GP-MAP fit,
    6 of 8 seeds reach the global optimum in 20 iterations and 5 of 8 within the reference demo's 15
    (profiles/r02_kat_1d.log; the parallel multi-start branch with its random starts: 3 of 8).  The misses sit in the local
    optimum x = 0.378 (f = 1.555): EI with a zero-mean GP and MAP hyper-parameters from a handful of points is
    over-confident there -- a property of the model, the reference's too."""
    def scan(iters, strategy):
        reached = 0
        for seed in range(1, 9):
            p = subprocess.run([os.path.join(BIN, "bayesian_optimization_1d"), "1", str(iters), str(seed)], capture_output=True,
                               text=True, timeout=600, env=dict(os.environ, SLS_GLOBAL_SEARCH=strategy))
            assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
            m = re.search(r"maximizer ([-\d.e]+) maximum ([-\d.e]+)", p.stdout)
            assert m, p.stdout
            assert float(m.group(2)) > 0.99                      # never below the boundary value f(0) = 1
            if abs(float(m.group(1)) - 0.852733) < 2e-2 and abs(float(m.group(2)) - 2.273928) < 2e-2:
                reached += 1
        return reached
    assert scan(20, "direct") >= 6
    assert scan(15, "direct") >= 4
    assert scan(20, "multistart") >= 2


def test_bayesian_optimization_2d_two_bump_scenario():
    """The two-bump objective and loop of the reference's 2-D GUI demo (demos/bayesian_optimization_2d_gui/core.cpp:26-55,80-88;
    SURVEY.md 4) as a CLI: GP MAP fit in four hyper-parameters + EI maximisation per iteration.  KAT (scipy.optimize from both
    bump centres): the sum has its maximum f = 1.532995 at x = (0.682333, 0.682333).  Every seed must be there within 25
    iterations (measured: 8 of 8 within 5e-3 in x and 1e-4 in f)."""
    for seed in range(1, 9):
        p = subprocess.run([os.path.join(BIN, "bayesian_optimization_2d"), "25", str(seed)], capture_output=True, text=True,
                           timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        m = re.search(r"maximizer ([-\d.e]+) ([-\d.e]+) maximum ([-\d.e]+)", p.stdout)
        assert m, p.stdout
        x0, x1, v = (float(m.group(i)) for i in (1, 2, 3))
        assert abs(x0 - 0.682333) < 0.02 and abs(x1 - 0.682333) < 0.02 and abs(v - 1.532995) < 2e-3, (seed, x0, x1, v)


def test_sequential_line_search_nd_demo():
    """Reference demo scenario (D = 8, 10 iterations): the residual to the optimum 0.4*1 trends down."""
    out = run("sequential_line_search_nd", 8, 10, 1)
    res = [float(x) for x in re.findall(r"residual ([-\d.e]+)", out)]
    assert len(res) == 10
    assert res[-1] < res[0] and res[-1] < 0.35, res


def test_sequential_line_search_nd_c3_size():
    """BASELINE config C3: sequential_line_search_nd at D = 32, 30 iterations -- full PreferenceRegressor MAP + EI acquisition
    per step through the C++ facade.  The synthetic user picks the best point of every slider, so the residual to the
    optimum 0.4*1 cannot grow by more than the slider's resolution and must trend down over the run."""
    out = run("sequential_line_search_nd", 32, 30, 1, timeout=900)
    res = [float(x) for x in re.findall(r"residual ([-\d.e]+)", out)]
    assert len(res) == 30
    assert res[-1] < res[0] and min(res[-5:]) < 0.8 * res[0], res
    assert all(np.isfinite(res))


def test_host_layer_environment_switches():
    """The host layer's process-wide settings (host/device.cpp; read once per process, hence one process each): the device(s) the lazily
    created contexts live on, NLopt's relative tolerances for the local searches (default 1e-6, 0 = run to the cap) and for the MAP
    fits (default off), the timing trace.  The scenario (D = 8, 8 iterations, the reference's use_MAP_hyperparams = true) must run under
    each and end where the default run ends to within what a polished local search can move the slider's end point."""
    def scenario(**env):
        p = subprocess.run([os.path.join(BIN, "sequential_line_search_nd"), "8", "8", "1", "1"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, **env))
        assert p.returncode == 0, (env, p.stdout[-2000:] + p.stderr[-2000:])
        res = [float(x) for x in re.findall(r"residual ([-\d.e]+)", p.stdout)]
        assert len(res) == 8 and all(np.isfinite(res)), (env, p.stdout[-2000:])
        return res, p.stderr
    base, err = scenario()
    assert "ms" not in err or "maximiser" not in err              # no timing trace unless asked for
    for env in (dict(SLS_DEVICE="0"), dict(SLS_DEVICES="0,0"), dict(SLS_LOCAL_SEARCH_TOL="0"), dict(SLS_LOCAL_SEARCH_TOL="1e-9"),
                dict(SLS_MAP_FIT_TOL="1e-6")):
        res, _ = scenario(**env)
        assert res[-1] < res[0], (env, res)
        if "SLS_DEVICE" in env:                                   # the same computation on the same device: the same numbers
            assert res == base, (env, res, base)
        else:
            assert abs(res[0] - base[0]) < 0.05 and res[-1] < 1.5 * base[-1] + 0.05, (env, res, base)
    _, err = scenario(SLS_HOST_TIMING="1")
    assert re.search(r"\d ms|ms \d|ms=", err), err[-1500:]        # stderr carries the per-call timing lines
    p = subprocess.run([os.path.join(BIN, "sequential_line_search_nd"), "8", "2", "1", "1"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, SLS_DEVICE="7"))
    assert p.returncode != 0 and ("device" in (p.stdout + p.stderr).lower())      # a device that is not there: a loud failure, no fallback
