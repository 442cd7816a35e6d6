"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/valley_hip.h declares (no compute calls without a GPU), and the product path refuses to run
without the HIP library / on CPU tensors."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "valley_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
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


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "valley_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
