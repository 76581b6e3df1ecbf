"""Kernel-by-kernel timeline of ONE fit at N = 8192, D = 64 (run under rocprofv3 --kernel-trace; tools/r06 scripts)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from util import sls, synth_problem
from oracle import oracle_py as oracle
m = sls()
ctx = m.Context(0)
X, y, theta, b = synth_problem(oracle, 64, 8192)
for _ in range(3):
    g = m.GP(ctx, X, y, theta, b, 1); g.close()
ctx.synchronize()
