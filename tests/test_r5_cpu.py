"""Round 5 host-side checks (no GPU).

The persistent GEMM's rolled instantiations own the accumulation registers BY NAME (valley_amd/csrc/gemm_bf16.hip, "accumulators by
name"): the compiler must never touch a0..a255 in them.  tools/agpr_audit.py reads the ISA hipcc emits and fails on any accumulation
register named outside an inline-asm block, on scratch and on spills; this test runs it on the fast build (the persistent tiles only)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_named_accumulators_are_never_touched_by_the_compiler():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "agpr_audit.py"), "-DVLY_FEW_TILES"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "12 kernels with accumulators by name audited, 0 findings" in r.stdout, r.stdout[-2000:]


def test_round6_kernels_with_named_accumulators_are_audited_too():
    """gemm_p32_kernel / gemm_p16_kernel (round 6) own a0 .. a255 the same way; valley_amd.build audits every library it compiles
    (valley_amd/agpr_audit.py) — this is the command-line form on both sources, bf16 and fp16 storage."""
    for args, n in ((["--p16"], 4), (["--p16", "-DVLY_FP16=1"], 4), (["--p32"], 8)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "agpr_audit.py"), *args], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        assert f"{n} kernels with accumulators by name audited, 0 findings" in r.stdout, r.stdout[-2000:]
