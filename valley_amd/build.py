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
    procs = []
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
            cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags, "-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
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
