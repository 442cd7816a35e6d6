"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/valley_hip.h declares (no compute calls without a GPU), and the product path refuses to run
without the HIP library / on CPU tensors."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(experimental: bool = False):
    """The entry points include/valley_hip.h declares: its default section (what the two shipped libraries export), or — with
    ``experimental`` — only the prototypes inside ``#ifdef VLY_EXPERIMENTAL`` blocks (libvalley_hip_exp.so adds those)."""
    txt = open(os.path.join(ROOT, "include", "valley_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    blocks = re.findall(r"#ifdef VLY_EXPERIMENTAL(.*?)#endif", txt, flags=re.S)
    if experimental:
        txt = "\n".join(blocks)
    else:
        txt = re.sub(r"#ifdef VLY_EXPERIMENTAL.*?#endif", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vly_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from valley_amd import build, lib
    build.build(verbose=False)
    names = header_symbols()
    assert len(names) >= 14
    handle = lib.load()
    for n in names:
        assert hasattr(handle, n), n
    assert sorted(lib.EXPORTS) == names            # the ctypes binding covers exactly the header
    assert handle.vly_abi_version() == lib.ABI_VERSION


def test_exported_symbols_are_exactly_the_header(tmp_path):
    """`nm -D` of the two shipped libraries = the header's default section = the ctypes binding (the libraries carry only
    what a default path calls); the experimental library adds exactly the header's EXPERIMENTAL prototypes."""
    import subprocess
    from valley_amd import build, lib
    build.build(verbose=False)

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return sorted(ln.split()[-1] for ln in out.splitlines() if re.search(r" T vly_[a-z0-9_]+$", ln))

    names, exp_names = header_symbols(), header_symbols(experimental=True)
    assert exported(build.LIB) == names == exported(build.LIB_F16)
    assert len(names) <= 52, len(names)          # (round 6: + vly_split3_f32, vly_norm_split3_f32)
    assert sorted(lib._SIGS_EXPERIMENTAL) == exp_names
    assert exported(build.LIB_EXP) == sorted(names + exp_names) == exported(build.LIB_EXP_F16)


def test_fp16_library_exports_the_same_abi():
    """libvalley_hip_f16.so (the same sources with -DVLY_FP16=1) exports every declared symbol and reports fp16 storage;
    the default library reports bf16 (host-side functions: no GPU needed)."""
    import ctypes
    from valley_amd import build, lib
    build.build(verbose=False)
    h16, hbf = ctypes.CDLL(build.LIB_F16), ctypes.CDLL(build.LIB)
    for n in header_symbols():
        assert hasattr(h16, n), n
    assert h16.vly_abi_version() == hbf.vly_abi_version() == lib.ABI_VERSION
    assert h16.vly_storage_dtype() == 1 and hbf.vly_storage_dtype() == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from valley_amd import lib
    monkeypatch.setattr(lib, "_LIB", None)
    monkeypatch.setenv("VALLEY_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(lib.ValleyHipError):
        lib.load()


def test_ops_reject_cpu_tensors():
    from valley_amd import lib, ops
    a = torch.zeros((16, 64), dtype=torch.bfloat16)
    with pytest.raises(lib.ValleyHipError):
        ops.gemm(a, a)
    with pytest.raises(lib.ValleyHipError):
        ops.rmsnorm(torch.zeros((4, 256)), torch.ones(256), 1e-5)


def test_packed_weight_needs_a_device_tensor():
    from valley_amd import lib, ops
    with pytest.raises(lib.ValleyHipError):
        ops.PackedWeight(torch.zeros((128, 64), dtype=torch.bfloat16))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "valley_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_shipped_tuning_table_is_consistent():
    """valley_amd/tuned/gfx950.json (decisions of the online GEMM tuner for the reference configurations) loads,
    names only kernels the dispatcher knows, and covers every GEMM shape of configs[1]."""
    import json
    import os

    from valley_amd import ops
    path = os.path.join(os.path.dirname(ops.__file__), "tuned", "gfx950.json")
    ents = json.load(open(path))
    assert len(ents) >= 11
    known = set(ops.CANDIDATES) | set(ops.SPLIT_CANDIDATES)
    for e in ents:
        assert (e["kind"], e["tile"]) in known, e
        assert e["key"][4] in ("torch.bfloat16", "torch.float32")
    keys = {tuple(e["key"][:4]) for e in ents}
    for shape in [(1312, 12288, 4096, 0), (1312, 4096, 4096, 0), (1312, 22016, 4096, 2), (1312, 4096, 11008, 0),
                  (8224, 3072, 1024, 0), (8224, 1024, 1024, 0), (8224, 4096, 1024, 1), (8224, 1024, 4096, 0)]:
        assert shape in keys, shape
    # the default engine reads the Llama projections from the block-ordered copy: configs[1]'s four prefill GEMMs are
    # decided for that layout too (8th key element), so that the default bench run does not start by tuning
    packed = {tuple(e["key"][:4]) for e in ents if e["key"][7:] == ["p64"]}
    for shape in [(1312, 12288, 4096, 0), (1312, 4096, 4096, 100), (1312, 22016, 4096, 2), (1312, 4096, 11008, 100)]:
        assert shape in packed, shape
    saved = dict(ops._TUNED)
    try:
        ops._TUNED.clear()
        assert ops.load_tune_cache(path) == len(ents)
        assert (1312, 22016, 4096, 2, "half", False, False, "p64") in ops._TUNED     # 16-bit storage types share the key "half"
    finally:
        ops._TUNED.clear()
        ops._TUNED.update(saved)
