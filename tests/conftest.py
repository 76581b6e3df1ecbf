import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """Evidence records of the GPU tests (tests/util.py: record): divergent-start counts, per-seed errors, timings."""
    try:
        import json

        from util import EVIDENCE
        if EVIDENCE:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "test_evidence.json"), "w") as f:
                json.dump(EVIDENCE, f, indent=1)
    except Exception:
        pass


@pytest.fixture(autouse=True)
def _sls_tuning(monkeypatch):
    """libsls_hip parses its SLS_* switches once per process (csrc/tuning.hpp); tests switch them with monkeypatch.setenv /
    delenv (or os.environ + util.tuning_reload()).  Here: every change made through monkeypatch re-reads them at once, and every
    test starts from the environment as it is (the previous test's changes are undone by then)."""
    from util import tuning_reload
    tuning_reload()
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_and_reload(name, value, *a, **k):
        setenv(name, value, *a, **k)
        if name.startswith("SLS_"):
            tuning_reload()

    def delenv_and_reload(name, *a, **k):
        delenv(name, *a, **k)
        if name.startswith("SLS_"):
            tuning_reload()
    monkeypatch.setenv, monkeypatch.delenv = setenv_and_reload, delenv_and_reload
    yield


@pytest.fixture(scope="session")
def fixtures():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "fixtures.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    # the oracle's OpenMP loops stop scaling (and on a 256-thread host slow down badly) past ~64 threads
    os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 64)))
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def scipy_cases():
    """tests/golden/scipy_pipelines.npz (make_scipy_fixtures.py): GP pipelines at N = 130 .. 2048 from scipy/LAPACK + numpy,
    as a dict  case name -> dict of arrays."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "scipy_pipelines.npz"))
    cases = {}
    for name in z["cases"]:
        name = str(name)
        cases[name] = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}
    cases["_ucb_h"] = float(z["ucb_h"])
    return cases
