"""The host C++ layer (reference public surface over the C ABI) on the GPU: behavioural checks + the two CLI scenarios."""
import os
import re
import subprocess

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
    """BASELINE config C1: 20 iterations; true optimum x = 0.852733, f = 2.273928."""
    out = run("bayesian_optimization_1d", 1, 20, 1)
    m = re.search(r"maximizer ([-\d.e]+) maximum ([-\d.e]+)", out)
    assert m, out
    assert abs(float(m.group(1)) - 0.852733) < 2e-2
    assert abs(float(m.group(2)) - 2.273928) < 2e-2


def test_sequential_line_search_nd_demo():
    """Reference demo scenario (D = 8, 10 iterations): the residual to the optimum 0.4*1 trends down."""
    out = run("sequential_line_search_nd", 8, 10, 1)
    res = [float(x) for x in re.findall(r"residual ([-\d.e]+)", out)]
    assert len(res) == 10
    assert res[-1] < res[0] and res[-1] < 0.35, res
