"""Audit of the kernels that own their accumulation registers BY NAME (gemm_p4_kernel's rolled instantiations, gemm_p32_kernel,
gemm_p16_kernel): in the ISA hipcc emits for them no instruction OUTSIDE an inline-asm block may name an accumulation register — the
compiler has no value there, and a spill or copy of its own into a0 .. a255 silently corrupts a tile.  Also required: no scratch.
valley_amd.build runs this on the ISA of EVERY library it builds (the fp16 library was shipped unaudited) and fails the
build on a finding; tools/agpr_audit.py is the command-line form."""
import re
import subprocess

AUDITED = {"gemm_bf16.hip": "gemm_p4_kernel", "gemm_p32.hip": "gemm_p32_kernel", "gemm_p16.hip": "gemm_p16_kernel"}


def audit_asm(asm_path, kern):
    """-> (lines of report, number of audited kernels, number of findings)"""
    asm = open(asm_path).read().splitlines()
    bad = kernels = 0
    report = []
    i = 0
    areg = re.compile(r"\ba(\[(\d+|0x[0-9a-f]+)(:(\d+|0x[0-9a-f]+))?\]|\d+\b)")
    while i < len(asm):
        m = re.match(r"^(_ZN\S*" + kern + r"\S*):", asm[i])
        if not m:
            i += 1
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
        j = i + 1
        in_asm = False
        outside, n_mfma, n_rd, n_mfma_compiler = [], 0, 0, 0
        while j < len(asm) and "s_endpgm" not in asm[j]:
            ln = asm[j]
            if ";;#ASMSTART" in ln:
                in_asm = True
            elif ";;#ASMEND" in ln:
                in_asm = False
            else:
                code = ln.split(";")[0]
                if in_asm:
                    if "v_mfma" in code:
                        n_mfma += 1
                    if "v_accvgpr_read" in code:
                        n_rd += 1
                elif areg.search(code) and not code.strip().startswith("."):
                    outside.append((j - i, code.strip()))
                    if "v_mfma" in code:
                        n_mfma_compiler += 1
                if "scratch_" in code:
                    outside.append((j - i, code.strip()))
            j += 1
        # the kernel descriptor must allocate every accumulation register the asm names (it does so only because of the clobber list
        # at the kernel's entry): next_free_vgpr - accum_offset >= highest named register + 1
        named = [int(x, 0) for ln in asm[i:j] for x in re.findall(r"\ba\[(?:\d+|0x[0-9a-f]+):(\d+|0x[0-9a-f]+)\]", ln.split(";")[0])]
        desc = "\n".join(asm[j:j + 120])
        nf = re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", desc)
        ao = re.search(r"\.amdhsa_accum_offset\s+(\d+)", desc)
        lit = n_mfma > 0 and n_mfma_compiler == 0 and "v_mfma" not in "\n".join(c for _, c in outside)
        if lit and named and nf and ao and int(nf.group(1)) - int(ao.group(1)) < max(named) + 1:
            outside.append((0, f"descriptor allocates {int(nf.group(1)) - int(ao.group(1))} accumulation registers, the asm names a{max(named)}"))
        if lit and not (nf and ao):
            outside.append((0, "kernel descriptor not found behind the kernel"))
        if n_mfma_compiler:              # the compiler manages this kernel's accumulators (builtin MFMAs): not audited
            lit = False
        if lit:
            kernels += 1
            status = "ok" if not outside else "FAIL"
            report.append(f"{name:48s} mfma {n_mfma:4d}  acc reads {n_rd:4d}  compiler AGPR / scratch instructions: {len(outside)}  {status}")
            for off, code in outside[:8]:
                report.append(f"    +{off}: {code}")
            bad += len(outside)
        i = j
    return report, kernels, bad
