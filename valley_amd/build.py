"""Builds libvalley_hip.so (gfx950) in-tree with hipcc.  No torch in the loop: the library is a
plain C-ABI shared object (include/valley_hip.h)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvalley_hip.so")
LIB_F16 = os.path.join(LIBDIR, "libvalley_hip_f16.so")      # the same sources with -DVLY_FP16=1 (IEEE fp16 storage)
# the bf16 library plus the EXPERIMENTAL entry points and kernels (include/valley_hip.h's last section): only the units below are
# compiled again, with -DVLY_EXPERIMENTAL=1; no default path loads it (VALLEY_EXPERIMENTAL=1 does; tests/test_experimental_gpu.py)
LIB_EXP = os.path.join(LIBDIR, "libvalley_hip_exp.so")
LIB_EXP_F16 = os.path.join(LIBDIR, "libvalley_hip_exp_f16.so")   # the same on fp16 storage (the experiments' bit-identity tests run on both types)
EXP_UNITS = ["decode_step.hip", "attention.hip", "gemm_bf16.hip", "gemv_bf16.hip"]
SOURCES = ["capi.hip", "gemm_bf16.hip", "gemm_p32.hip", "gemm_p16.hip", "gemm_streamk.hip", "norm_elementwise.hip", "attention.hip", "temporal_delta.hip", "preprocess.hip", "gemv_bf16.hip", "precise_f32.hip", "gemm_skinny.hip", "decode_step.hip"]


MAX_PARALLEL = max(2, min(8, (os.cpu_count() or 4)))     # hipcc processes at once (a clean build used to launch ~30: ADVICE r5)


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "valley_hip.h")]
    for lib in (LIB, LIB_F16, LIB_EXP, LIB_EXP_F16):
        if not os.path.exists(lib):
            return True
        t = os.path.getmtime(lib)
        if any(os.path.getmtime(d) > t for d in deps):
            return True
    return False


def build(force: bool = False, verbose: bool = True) -> str:
    """The two shipped libraries (bf16 and fp16 storage) and the experimental one, every stale translation unit compiled in parallel."""
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(os.path.join(LIBDIR, "f16"), exist_ok=True)
    os.makedirs(os.path.join(LIBDIR, "exp"), exist_ok=True)
    os.makedirs(os.path.join(LIBDIR, "exp_f16"), exist_ok=True)
    if not force and not needs_build():
        return LIB
    variants = [(LIB, LIBDIR, [], SOURCES), (LIB_F16, os.path.join(LIBDIR, "f16"), ["-DVLY_FP16=1"], SOURCES),
                (LIB_EXP, os.path.join(LIBDIR, "exp"), ["-DVLY_EXPERIMENTAL=1"], EXP_UNITS),
                (LIB_EXP_F16, os.path.join(LIBDIR, "exp_f16"), ["-DVLY_EXPERIMENTAL=1", "-DVLY_FP16=1"], EXP_UNITS)]
    try:
        from .agpr_audit import AUDITED, audit_asm
    except ImportError:                                                   # (run as a script: python valley_amd/build.py)
        sys.path.insert(0, HERE)
        from agpr_audit import AUDITED, audit_asm
    jobs, audits = [], []
    # a translation unit is recompiled when its own source, a shared header (*.hpp, *.inc, valley_hip.h) or this recipe is newer than
    # its object — editing one kernel file costs one compile per library, not twenty-two
    shared = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith(".hip")] + \
             [os.path.join(HERE, "..", "include", "valley_hip.h"), os.path.abspath(__file__)]
    t_shared = max(os.path.getmtime(d) for d in shared)
    for lib, odir, flags, units in variants:
        for s in units:
            o = os.path.join(odir, s.replace(".hip", ".o"))
            if not force and os.path.exists(o) and os.path.getmtime(o) >= max(t_shared, os.path.getmtime(os.path.join(CSRC, s))):
                continue
            # the units whose kernels hold accumulation registers by name keep their ISA (-save-temps=obj) for the audit below
            temps = ["-save-temps=obj"] if s in AUDITED else []
            cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags, *temps, "-c", os.path.join(CSRC, s), "-o", o]
            jobs.append((s, cmd))
            if s in AUDITED:
                audits.append((s, odir, " ".join(flags) or "(default flags)"))
    running = []
    while jobs or running:
        while jobs and len(running) < MAX_PARALLEL:
            s, cmd = jobs.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        s, p = running.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    # every library that was (re)compiled: no compiler-made access to the accumulation registers the kernels own by name
    for s, odir, what in audits:
        stem = os.path.join(odir, s[:-4])
        asm = stem + "-hip-amdgcn-amd-amdhsa-gfx950.s"
        report, kernels, bad = audit_asm(asm, AUDITED[s])
        for ext in ("-hip-amdgcn-amd-amdhsa-gfx950.s", "-hip-amdgcn-amd-amdhsa-gfx950.bc", "-hip-amdgcn-amd-amdhsa-gfx950.hipi",
                    "-hip-amdgcn-amd-amdhsa-gfx950.o", "-hip-amdgcn-amd-amdhsa-gfx950.out", "-hip-amdgcn-amd-amdhsa-gfx950.out.resolution.txt",
                    "-host-x86_64-unknown-linux-gnu.bc", "-host-x86_64-unknown-linux-gnu.hipi", "-host-x86_64-unknown-linux-gnu.s",
                    ".hip-hip-amdgcn-amd-amdhsa.hipfb"):
            try:
                os.remove(stem + ext)
            except OSError:
                pass
        if verbose:
            print(f"agpr audit {s} [{what}]: {kernels} kernels by name, {bad} findings")
        if bad or not kernels:
            os.remove(os.path.join(odir, s.replace(".hip", ".o")))         # (never link, never ship, recompile next time)
            raise RuntimeError(f"accumulation-register audit failed for {s} [{what}]:\n" + "\n".join(report))
    for lib, odir, _flags, units in variants:
        base = os.path.join(LIBDIR, "f16") if "-DVLY_FP16=1" in _flags else LIBDIR      # the units a variant does not recompile
        objs = [os.path.join(odir if s in units else base, s.replace(".hip", ".o")) for s in SOURCES]
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


ABI_SMOKE_SRC = os.path.join(HERE, "..", "tests", "c_abi", "abi_smoke.cpp")
ABI_SMOKE = os.path.join(LIBDIR, "abi_smoke")


def build_abi_smoke(verbose: bool = True) -> str:
    """The torch-free C++ host program of tests/c_abi (dlopens the library; tests/test_kernels_gpu.py runs it)."""
    hdr = os.path.join(HERE, "..", "include", "valley_hip.h")          # VLY_ABI_VERSION is compiled in
    if os.path.exists(ABI_SMOKE) and os.path.getmtime(ABI_SMOKE) >= max(os.path.getmtime(ABI_SMOKE_SRC), os.path.getmtime(hdr)):
        return ABI_SMOKE
    cmd = [hipcc(), "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", ABI_SMOKE_SRC, "-o", ABI_SMOKE, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return ABI_SMOKE


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
