"""The batched DIRECT of the host layer (host/direct.cpp; the global phase of the reference's default maximiser branch,
src/acquisition-function.cpp:155-165) walks a pinned trajectory: tools/probes/direct_bench.cpp hashes every point it evaluates on a
synthetic objective, and the hashes below were recorded with the implementation of rounds 2-3 (one heap object per rectangle)
before its bookkeeping was rewritten structure-of-arrays in round 4.  CPU only: no device call is involved."""
import os, re, subprocess
import pytest

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINNED = {   # (D, max_evals) -> (evaluations used, hash over the evaluated points of one run + result + best value)
    (32, 1600): (1581, "d73cc7c6c18c0a7b"),
    (1, 50): (49, "d69a56d65b6e00dc"),
    (3, 300): (297, "9241ce2c06f02e1a"),
    (7, 700): (695, "f23ee589fb95d240"),
    (130, 300): (261, "53fc79f92bbae955"),
}


@pytest.fixture(scope="module")
def bench(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("direct") / "direct_bench")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(R, "include"), "-I" + os.path.join(R, "sequential-line-search_amd", "host"),
                    os.path.join(R, "tools", "probes", "direct_bench.cpp"), os.path.join(R, "sequential-line-search_amd", "host", "direct.cpp"),
                    "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("D,evals", sorted(PINNED))
def test_direct_trajectory_is_pinned(bench, D, evals):
    out = subprocess.run([bench, str(D), str(evals), "3"], check=True, capture_output=True, text=True).stdout
    m = re.search(r"used (\d+),.*hash ([0-9a-f]{16})", out)
    assert m, out
    # three runs in one process, the hash is the last run's: the storage kept between calls must not leak state into the next run
    assert (int(m.group(1)), m.group(2)) == PINNED[(D, evals)], out
