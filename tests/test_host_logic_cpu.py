"""Host logic of the C++ layer that needs no GPU (data manager, slider, BTL, bounded L-BFGS driver, CSV round trip,
kernel scalar forms) and its error behaviour when no device is present.  Runs in the CPU-only container."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_binary():
    import torch

    import __graft_entry__ as g
    g.build()
    exe = os.path.join(ROOT, "sequential-line-search_amd", "bin", "test_host_cpu")
    args = [exe] if torch.cuda.is_available() else [exe, "--no-gpu"]
    p = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "HOST CPU TESTS PASSED" in p.stdout, p.stdout + p.stderr
