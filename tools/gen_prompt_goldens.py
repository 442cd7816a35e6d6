#!/usr/bin/env python3
"""Capture the reference's prompt glue (valley_model.py:381-422: build_inputs, process_response) on a fake
tokenizer -> tests/golden/g6_prompt.json.  Authoring container only (imports /root/reference)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.fake_tokenizer import SPECIALS, FakeTokenizer  # noqa: E402
from tools.gen_goldens import import_reference  # noqa: E402

vm = import_reference()
cls = vm.ValleyLlamaForCausalLM
tok = FakeTokenizer()
tok.add_tokens(SPECIALS, special_tokens=True)
cases = {
    "video": [{"role": "system", "content": "You are Valley."}, {"role": "user", "content": "Describe this video concisely.\n<video>"}],
    "image_multi_turn": [{"role": "system", "content": "sys"}, {"role": "user", "content": "what is in <image> ?"},
                         {"role": "assistent", "content": "a cat"}, {"role": "user", "content": "and here <video> now"}],
}
out = {"build_inputs": {}, "errors": {}, "process_response": {}}
for k, m in cases.items():
    out["build_inputs"][k] = cls.build_inputs(None, tok, m).input_ids
for k, m in {"no_video": [{"role": "user", "content": "hello"}], "bad_role": [{"role": "robot", "content": "<video>"}]}.items():
    try:
        cls.build_inputs(None, tok, m)
        out["errors"][k] = "no error"
    except Exception as e:  # noqa: BLE001
        out["errors"][k] = f"{type(e).__name__}: {e}"
resp = ["### Assistant: a dog runs ### Human: next", "Valley: ###Response: hi there", "no separator at all", "  ###  ### x ###"]
out["process_response"] = {"inputs": resp, "outputs": cls.process_response(None, resp)}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "g6_prompt.json"), "w"))
print({k: len(v[0]) for k, v in out["build_inputs"].items()}, out["errors"], out["process_response"]["outputs"])
