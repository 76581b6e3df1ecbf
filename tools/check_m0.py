#!/usr/bin/env python3
"""Every compiler-placed reader of M0 must see an M0 written by the compiler in the same basic block, with no inline-asm
statement in between: gemm_f64.hpp writes M0 inside asm statements (declared as clobbered, but M0 is a reserved register and
clang does not promise to honour that).  Compiles the kernel sources to ISA and checks it.  usage: python tools/check_m0.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sequential-line-search_amd", "csrc")
READS_M0 = re.compile(r"^\s+(global_load_lds|buffer_load.*\blds\b|ds_gws|s_movrel|v_movrel|v_interp|s_sendmsg|v_readlane_b32 .*m0|v_writelane_b32 .*m0)")
WRITES_M0 = re.compile(r"^\s+s_\w+\s+m0,")


def check(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "-mllvm",
                        "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only", "-o", out, src], check=True,
                       stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    bad = n = 0
    in_asm = False
    for i, line in enumerate(lines):
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif not in_asm and READS_M0.match(line):
            n += 1
            j = i - 1
            while j >= 0:
                p = lines[j]
                if WRITES_M0.match(p):
                    break
                if "#ASMEND" in p:                                # an asm statement: harmless unless it touches M0
                    k = j
                    while k >= 0 and "#ASMSTART" not in lines[k]:
                        k -= 1
                    if any("m0" in q for q in lines[k:j]):
                        bad += 1
                        print(f"{os.path.basename(src)}:{i + 1}: an asm statement writes M0 between the compiler's write and: {line.strip()}")
                        break
                    j = k
                elif re.match(r"^[.\w$]+:", p):                   # block boundary first
                    bad += 1
                    print(f"{os.path.basename(src)}:{i + 1}: M0 read without a compiler write in its block: {line.strip()}")
                    break
                j -= 1
    return n, bad


def main():
    total = bad = 0
    for f in sorted(os.listdir(CSRC)):
        if f.startswith("kernels_") and f.endswith(".hip"):
            n, b = check(os.path.join(CSRC, f))
            print(f"{f}: {n} compiler-placed M0 readers, {b} suspicious")
            total += n
            bad += b
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
