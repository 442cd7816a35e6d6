"""The EXPERIMENTAL library (libvalley_hip_exp.so = the bf16 sources with -DVLY_EXPERIMENTAL=1, valley_amd/build.py): kernels and
entry points that were built to parity and measured behind the default path — the persistent decode step (vly_decode_layers),
tile hint 297 of vly_gemm_bf16_streamk, round 3's split / merge form of the decode attention, the witness attention kernels.  No default path loads that library; a process selects it with
VALLEY_EXPERIMENTAL=1, so its tests run in a child here: the whole of tests/test_decode_persistent_gpu.py and tests/test_decode_merge_gpu.py, the hint-297 and split-and-merge cases of
tests/test_kernels_gpu.py (all skip themselves on the shipped library)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_experimental_library_suite():
    env = dict(os.environ, VALLEY_EXPERIMENTAL="1")
    env.pop("VALLEY_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_decode_persistent_gpu.py"),
                        os.path.join(ROOT, "tests", "test_decode_merge_gpu.py"), os.path.join(ROOT, "tests", "test_kernels_gpu.py"),
                        "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "persistent_gpu or decode_merge_gpu or (p4_streamk and 297) or split_and_merge"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = r.stdout[-2500:]
    assert r.returncode == 0, tail + r.stderr[-1500:]
    assert " passed" in tail and "skipped" not in tail.splitlines()[-1], tail


def test_shipped_library_has_no_experimental_exports():
    import ctypes
    from valley_amd import build, lib
    shipped = ctypes.CDLL(build.LIB)                        # (whatever VALLEY_EXPERIMENTAL says about the library this process loaded)
    for name in lib._SIGS_EXPERIMENTAL:
        assert not hasattr(shipped, name), name
    exp = ctypes.CDLL(build.LIB_EXP)
    for name in list(lib._SIGS_EXPERIMENTAL) + list(lib.EXPORTS):
        assert hasattr(exp, name), name
