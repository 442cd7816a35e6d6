"""Round 4 host logic (no GPU): the caller's dtype decides which library a process binds.

Reference behaviour being matched: ``ValleyLlamaForCausalLM.from_pretrained(model_name, torch_dtype=torch.float16)``
(/root/reference/valley/inference/run_valley.py:39, valley/serve/model_worker.py:61,79) and ``.half()`` on the frames
(valley/model/valley_model.py:430) run the model in IEEE fp16.  Here fp16 storage is a second library
(libvalley_hip_f16.so); a request for it must select that library while the choice is open and RAISE once the process is bound
to the other one — never run in bf16 silently (VERDICT r3 missing #4)."""
import os
import subprocess
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(code: str, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("VALLEY_PRECISION", "VALLEY_HIP_LIB")}
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code], env=e, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_fp16_request_selects_the_fp16_library_while_the_choice_is_open():
    out = _child("""
import torch
from valley_amd import runtime, lib, valley_model as vm
print("before", runtime.PRECISION, runtime.half_bound(), lib.lib_path().endswith("libvalley_hip.so"))
cfg = vm.ValleyConfig(vocab_size=8, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=1)
vm.apply_torch_dtype(cfg, torch.float16, "from_pretrained(torch_dtype=torch.float16)")
print("after", runtime.PRECISION, runtime.HALF, cfg.valley_precision, runtime.half_bound(), lib.lib_path().endswith("libvalley_hip_f16.so"))
l = lib.load()                      # the library loads without a GPU; it must be the fp16 one
print("storage", l.vly_storage_dtype())
try:
    vm.apply_torch_dtype(cfg, torch.bfloat16, "second request")
except ValueError as e:
    print("raised", "bound to fp16" in str(e))
""")
    assert "before bf16 False True" in out
    assert "after fp16 torch.float16 fp16 True True" in out
    assert "storage 1" in out and "raised True" in out


def test_env_pins_the_library_and_a_conflicting_request_raises():
    out = _child("""
import torch
from valley_amd import runtime, valley_model as vm
cfg = vm.ValleyConfig(vocab_size=8, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=1)
print("bound", runtime.half_bound())
vm.apply_torch_dtype(cfg, torch.bfloat16, "same type")          # what the env says: fine
print("ok", cfg.valley_precision)
for req in (torch.float16, "config"):
    try:
        if req == "config":
            cfg.valley_precision = "fp16"
            vm.resolve_precision(cfg)
        else:
            vm.apply_torch_dtype(cfg, req, "from_pretrained(torch_dtype=torch.float16)")
        print("no error")
    except ValueError as e:
        print("raised", "VALLEY_PRECISION" in str(e))
""", env={"VALLEY_PRECISION": "bf16"})
    assert "bound True" in out and "ok bf16" in out and out.count("raised True") == 2 and "no error" not in out


def test_loaded_library_binds_the_type():
    out = _child("""
import torch
from valley_amd import runtime, lib
lib.load()
print("bound", runtime.half_bound(), runtime.HALF)
try:
    runtime.request_half(torch.float16, ".half()")
except ValueError as e:
    print("raised", "lib.load(libvalley_hip.so)" in str(e))     # the loader names itself as what bound the type
runtime.request_half(torch.bfloat16, "same")                    # the bound type: no-op
runtime.request_half(torch.float32, "not a 16-bit request")     # ignored here
print("done")
""")
    assert "bound True torch.bfloat16" in out and "raised True" in out and "done" in out


def test_fp32_request_selects_the_validation_engines_and_model_dtype_checks():
    from valley_amd import valley_model as vm
    cfg = vm.ValleyConfig(vocab_size=8, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=1)
    vm.apply_torch_dtype(cfg, torch.float32, "from_pretrained(torch_dtype=torch.float32)")
    assert cfg.valley_precision == "fp32"
    vm.apply_torch_dtype(cfg, None, "no request")
    assert cfg.valley_precision == "fp32"
    with pytest.raises(ValueError):
        vm.apply_torch_dtype(cfg, torch.float64, "unsupported")
    # .to(dtype) / .half() on a built model: a matching dtype is a no-op, another one raises (no device needed for the check)
    m = vm.ValleyLlamaForCausalLM.__new__(vm.ValleyLlamaForCausalLM)
    m.device = torch.device("cuda:0")
    m.model = types.SimpleNamespace(wdtype=torch.bfloat16)
    assert m.to(torch.bfloat16) is m and m.to("cuda:0") is m and m.bfloat16() is m and m.dtype == torch.bfloat16
    for bad in (lambda: m.half(), lambda: m.to(torch.float16), lambda: m.to(dtype=torch.float32), lambda: m.float()):
        with pytest.raises(ValueError, match="storage type is fixed"):
            bad()
    m.model.wdtype = torch.float16
    assert m.half() is m and m.to(torch.float16) is m


def test_cli_entry_dtype_follows_the_reference_unless_bound():
    out = _child("""
import torch
from valley_amd import cli, runtime
print("open", cli.entry_dtype())
runtime.request_half(torch.bfloat16, "someone")
print("bound", cli.entry_dtype())
""")
    assert "open torch.float16" in out and "bound torch.bfloat16" in out


def test_cli_entry_dtype_keeps_the_fp32_validation_mode():
    """with VALLEY_PRECISION=fp32 the entry points must hand from_pretrained torch.float32 — a 16-bit dtype would let
    apply_torch_dtype overwrite config.valley_precision with "bf16" and silently build the bf16 engines."""
    out = _child("""
import torch
from valley_amd import cli, runtime, valley_model as vm
print("entry", cli.entry_dtype())
cfg = vm.ValleyConfig(vocab_size=8, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=1)
vm.apply_torch_dtype(cfg, cli.entry_dtype(), "test")
print("precision", cfg.valley_precision, runtime.PRECISION)
""", env={"VALLEY_PRECISION": "fp32"})
    assert "entry torch.float32" in out and "precision fp32 fp32" in out


def test_engine_constructors_bind_the_storage_type():
    """whatever allocates the first 16-bit tensor fixes the storage type — a later request for the other one raises
    instead of leaving mixed-dtype engines behind."""
    out = _child("""
import torch
from valley_amd import runtime
from valley_amd.vision_tower import HipCLIPVisionTower
print("open", runtime.half_bound())
HipCLIPVisionTower(device="cpu")
print("bound", runtime.half_bound())
try:
    runtime.request_half(torch.float16, "late caller")
    print("no error")
except ValueError as e:
    print("raised", "HipCLIPVisionTower" in str(e))
""")
    assert "open False" in out and "bound True" in out and "raised True" in out
