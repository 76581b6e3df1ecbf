"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/sls_hip.h declares,
and refuses to run without a GPU (no silent CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from util import sls

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sls_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sls_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    m = sls()
    lib = m.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libsls_hip.so does not export {n}"
    assert sorted(m.EXPORTS) == names
    assert lib.sls_version() >= 100


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = sls()
    with pytest.raises(m.SlsError) as e:
        m.Context(0)
    assert "no CPU fallback" in str(e.value) or "no HIP device" in str(e.value)


def test_merge_rank_results_first_maximum():
    m = sls()
    best = m.merge_rank_results([(1.0, 7, [0.1]), (2.0, 40, [0.2]), (2.0, 12, [0.3]), (-1.0, 0, [0.4])])
    assert best[0] == 2.0 and best[1] == 12 and best[2][0] == 0.3


def test_m0_is_not_live_across_the_saddr_load_statements():
    """gemm_f64.hpp writes M0 inside its inline-assembly LDS-direct loads (the compiler only emits the SADDR form outside loops)
    and lists it as clobbered, which clang does not promise to honour for a reserved register.  tools/check_m0.py compiles the
    kernel sources to gfx950 ISA and verifies that every compiler-placed reader of M0 sees a compiler-placed write in its own
    basic block with none of those statements in between."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_m0.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "kernels_acq.hip" in r.stdout and "kernels_chol.hip" in r.stdout


def test_boundary_header_is_plain_c_and_links(tmp_path):
    """The drop-in boundary is a C ABI: include/sls_hip.h must compile as C99 with -pedantic (no C++ in the signatures, no torch types),
    and a plain C program that calls it must link against libsls_hip.so and -- here, without a GPU -- get the library's error code
    and message back instead of a crash."""
    import subprocess
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include "sls_hip.h"\n'
                   'int main(void) {\n'
                   '    sls_ctx* c = 0;\n'
                   '    int rc = sls_ctx_create(0, &c);\n'
                   '    printf("version %d rc %d msg %s\\n", sls_version(), rc, rc ? sls_last_error() : "ok");\n'
                   '    if (c) sls_ctx_destroy(c);\n'
                   '    return 0;\n}\n')
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "sequential-line-search_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe = tmp_path / "probe"
    r = subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lsls_hip", "-Wl,-rpath," + libdir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and re.search(r"version \d+ rc -?\d+ msg ", r.stdout), r.stdout + r.stderr
    import torch
    if not torch.cuda.is_available():
        assert " rc 0 " not in r.stdout and ("no HIP device" in r.stdout or "no CPU fallback" in r.stdout), r.stdout
