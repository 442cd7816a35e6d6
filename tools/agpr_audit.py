#!/usr/bin/env python3
"""Command-line form of valley_amd/agpr_audit.py (the build runs the same audit on every library it produces): compiles one of the
by-name kernels' sources with the given flags and audits the ISA.
Usage: agpr_audit.py [--p32 | --p16] [-Dflags ...]   (exit code 1 on any finding)"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from valley_amd.agpr_audit import AUDITED, audit_asm  # noqa: E402


def main():
    flags = sys.argv[1:]
    unit = "gemm_bf16.hip"
    for opt, u in (("--p32", "gemm_p32.hip"), ("--p16", "gemm_p16.hip")):
        if opt in flags:
            flags.remove(opt)
            unit = u
    src = os.path.join(ROOT, "valley_amd", "csrc", unit)
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags, "-save-temps", "-c", src,
                               "-o", os.path.join(td, "g.o")], cwd=td, stderr=subprocess.DEVNULL)
        report, kernels, bad = audit_asm(os.path.join(td, unit[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s"), AUDITED[unit])
    print("\n".join(report))
    print(f"{kernels} kernels with accumulators by name audited, {bad} findings")
    return 1 if bad or not kernels else 0


if __name__ == "__main__":
    sys.exit(main())
