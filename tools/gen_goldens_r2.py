#!/usr/bin/env python3
"""Round-2 golden vectors, captured from the REFERENCE / the HF sub-modules it calls (authoring container only).

Adds to tools/gen_goldens.py's set (which is left untouched: its 11 fixtures regenerate bit-identically):

  g5_decode2.npz   prefill + 8 greedy KV-decode steps (valley/serve/model_worker.py:371-394 loop on the reference model) on a
                   prompt whose greedy tokens are not a repeat and whose top-2 logit gaps are all > 0.2
  g6_ops.npz       op-level outputs of the HF sub-modules the reference's hot path executes: LlamaRMSNorm
                   at eps 1e-5 and 1e-6, rotary embedding + apply_rotary_pos_emb at positions {0, 1, 327, 2047},
                   LlamaAttention (eager, causal, left-padded batch, head_dim 128), LlamaMLP (SwiGLU), nn.LayerNorm 1024,
                   CLIPMLP (quick_gelu), CLIPAttention (N = 257, head_dim 64)
  g8_apply_delta.npz   outputs of the reference's own valley/model/apply_delta.py:apply_delta on tiny checkpoints
  g9_conversation.json prompts of the reference's conversation templates after a scripted exchange

Inputs and weights are regenerated from (seed, name, shape) by the tests (valley_amd.weights); fixtures hold outputs only.
Usage: python tools/gen_goldens_r2.py
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import golden_cfg as G  # noqa: E402
from tests.golden_r2_cfg import OPS, delta_states, op_input, op_weights  # noqa: E402
from tools.gen_goldens import build_reference, import_reference, run  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def gen_decode2(vm):
    model = build_reference(vm, "mean")
    T = G.GCFG["T"]
    ids, _ = G.golden_ids("decode2")
    img1 = t(G.golden_pixels(T, "mixed")).view(1, T, 3, 224, 224)
    out, _ = run(model, t(ids), img1)
    past, logits = out.past_key_values, out.logits
    toks, lasts = [], []
    for _ in range(8):
        last = logits[:, -1, :]
        lasts.append(last.numpy().copy())
        token = last.argmax(-1)
        toks.append(token.numpy().copy())
        ctx = past.get_seq_length()
        with torch.no_grad():
            o = model(input_ids=token[:, None], use_cache=True, attention_mask=torch.ones(1, ctx + 1, dtype=torch.long),
                      past_key_values=past)
        logits, past = o.logits, o.past_key_values
    toks, lasts = np.stack(toks, 1), np.stack(lasts, 1)
    srt = np.sort(lasts[0], -1)
    print("decode2 tokens", toks, "min top-2 gap", float((srt[:, -1] - srt[:, -2]).min()))
    np.savez_compressed(os.path.join(GOLD, "g5_decode2.npz"), ids=ids, prefill_last=out.logits.numpy()[:, -1].copy(),
                        tokens=toks, last_logits=lasts)


def gen_ops():
    from transformers import CLIPVisionConfig, LlamaConfig
    from transformers.models.clip import modeling_clip as MC
    from transformers.models.llama import modeling_llama as ML
    out = {}
    o = OPS
    with torch.no_grad():
        # ---- LlamaRMSNorm (hf:llama/modeling_llama.py:51-67)
        x = t(op_input("rms.x", (5, o["H"])))
        for eps in (1e-5, 1e-6):
            m = ML.LlamaRMSNorm(o["H"], eps=eps)
            m.weight.copy_(t(op_weights("rms.w", (o["H"],), 0.1, 1.0)))
            out[f"rmsnorm_eps{eps:g}"] = m(x).numpy()
        # ---- rotary embedding at chosen positions (:73-124, 127-157)
        lcfg = LlamaConfig(hidden_size=o["H"], num_attention_heads=o["heads"], num_key_value_heads=o["heads"],
                           intermediate_size=o["I"], num_hidden_layers=1, vocab_size=32, max_position_embeddings=2048,
                           rms_norm_eps=1e-6, attn_implementation="eager")
        rot = ML.LlamaRotaryEmbedding(lcfg)
        pos = torch.tensor([o["rope_positions"]])
        q = t(op_input("rope.q", (1, o["heads"], len(o["rope_positions"]), 128)))
        k = t(op_input("rope.k", (1, o["heads"], len(o["rope_positions"]), 128)))
        cos, sin = rot(q, pos)
        qr, kr = ML.apply_rotary_pos_emb(q, k, cos, sin)
        out["rope_q"], out["rope_k"], out["rope_cos"], out["rope_sin"] = qr.numpy(), kr.numpy(), cos.numpy(), sin.numpy()
        # ---- LlamaAttention, eager, causal + left padding (:191-289)
        att = ML.LlamaAttention(lcfg, 0).eval()
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            getattr(att, n).weight.copy_(t(op_weights(f"att.{n}", (o["H"], o["H"]), 0.05)))
        B, S = 2, o["S"]
        h = t(op_input("att.h", (B, S, o["H"])))
        am = torch.ones((B, S), dtype=torch.long)
        am[1, :o["pad"]] = 0
        cos, sin = rot(h, torch.arange(S)[None].expand(B, S))
        i, j = torch.arange(S)[:, None], torch.arange(S)[None, :]
        allowed = (j <= i)[None, None] & am[:, None, None, :].bool()
        mask = torch.where(allowed, 0.0, torch.finfo(torch.float32).min)
        y, _ = att(h, (cos, sin), mask)
        out["llama_attention"] = y.numpy()
        # ---- LlamaMLP (SwiGLU, :160-173)
        mlp = ML.LlamaMLP(lcfg)
        mlp.gate_proj.weight.copy_(t(op_weights("mlp.gate", (o["I"], o["H"]), 0.05)))
        mlp.up_proj.weight.copy_(t(op_weights("mlp.up", (o["I"], o["H"]), 0.05)))
        mlp.down_proj.weight.copy_(t(op_weights("mlp.down", (o["H"], o["I"]), 0.05)))
        out["llama_mlp"] = mlp(t(op_input("mlp.x", (7, o["H"])))).numpy()
        # ---- CLIP pieces (hf:clip/modeling_clip.py:259-350)
        ccfg = CLIPVisionConfig(hidden_size=1024, num_attention_heads=16, intermediate_size=o["VI"], num_hidden_layers=1,
                                image_size=224, patch_size=14, attn_implementation="eager")
        ln = torch.nn.LayerNorm(1024, eps=ccfg.layer_norm_eps)
        ln.weight.copy_(t(op_weights("ln.w", (1024,), 0.1, 1.0)))
        ln.bias.copy_(t(op_weights("ln.b", (1024,), 0.1)))
        out["layernorm"] = ln(t(op_input("ln.x", (5, 1024)))).numpy()
        cm = MC.CLIPMLP(ccfg)
        cm.fc1.weight.copy_(t(op_weights("cmlp.fc1.w", (o["VI"], 1024), 0.03)))
        cm.fc1.bias.copy_(t(op_weights("cmlp.fc1.b", (o["VI"],), 0.1)))
        cm.fc2.weight.copy_(t(op_weights("cmlp.fc2.w", (1024, o["VI"]), 0.03)))
        cm.fc2.bias.copy_(t(op_weights("cmlp.fc2.b", (1024,), 0.1)))
        out["clip_mlp"] = cm(t(op_input("cmlp.x", (9, 1024)))).numpy()
        ca = MC.CLIPAttention(ccfg).eval()
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            getattr(ca, n).weight.copy_(t(op_weights(f"catt.{n}.w", (1024, 1024), 0.03)))
            getattr(ca, n).bias.copy_(t(op_weights(f"catt.{n}.b", (1024,), 0.1)))
        y, _ = ca(t(op_input("catt.x", (2, 257, 1024))))
        out["clip_attention"] = y.numpy()[:, ::4].copy()
    np.savez_compressed(os.path.join(GOLD, "g6_ops.npz"), **out)
    print("g6_ops:", {k: v.shape for k, v in out.items()})


def gen_apply_delta(vm):
    """Run the reference's apply_delta (valley/model/apply_delta.py:13-37) on tiny saved checkpoints."""
    from transformers import AutoTokenizer, LlamaConfig, LlamaForCausalLM  # noqa: F401
    base_sd, delta_sd, dims = delta_states()
    # `from valley import ValleyLlamaForCausalLM` (apply_delta.py:10) has no provider in the reference tree: give the
    # name to the module object python already holds for the reference's namespace package
    import valley
    valley.ValleyLlamaForCausalLM = vm.ValleyLlamaForCausalLM
    saved = list(sys.path)
    sys.path[:] = ["/root/reference"] + [p for p in saved if os.path.abspath(p or ".") != ROOT]
    try:
        import valley.model.apply_delta as AD
    finally:
        sys.path[:] = saved
    with tempfile.TemporaryDirectory() as tmp:
        bdir, ddir, tdir = (os.path.join(tmp, n) for n in ("base", "delta", "target"))
        bcfg = LlamaConfig(vocab_size=dims["vocab_base"], hidden_size=dims["H"], intermediate_size=dims["I"],
                           num_hidden_layers=dims["L"], num_attention_heads=dims["heads"], num_key_value_heads=dims["heads"],
                           tie_word_embeddings=False)
        base = LlamaForCausalLM(bcfg)
        missing, unexpected = base.load_state_dict({k: t(v) for k, v in base_sd.items()}, strict=False)
        assert not unexpected and all("rotary" in m for m in missing), (missing, unexpected)
        base.save_pretrained(bdir)
        dcfg = vm.ValleyConfig(vocab_size=dims["vocab"], hidden_size=dims["H"], intermediate_size=dims["I"],
                               num_hidden_layers=dims["L"], num_attention_heads=dims["heads"],
                               num_key_value_heads=dims["heads"], tie_word_embeddings=False)
        dcfg.use_mm_proj, dcfg.mm_hidden_size = True, 1024
        delta = vm.ValleyLlamaForCausalLM(dcfg)
        missing, unexpected = delta.load_state_dict({k: t(v) for k, v in delta_sd.items()}, strict=False)
        assert not unexpected and all("rotary" in m for m in missing), (missing, unexpected)
        delta.save_pretrained(ddir)
        # apply_delta also copies the tokenizer (:21,35): hand it a stub with the two methods it uses
        class _Tok:
            def save_pretrained(self, p):
                pass
        AD.AutoTokenizer = types.SimpleNamespace(from_pretrained=lambda p: _Tok())
        # its loaders pass torch_dtype=float16; run the arithmetic in fp32 so the fixture is exact
        orig_b, orig_d = AD.AutoModelForCausalLM.from_pretrained, AD.ValleyLlamaForCausalLM.from_pretrained
        AD.AutoModelForCausalLM = types.SimpleNamespace(from_pretrained=lambda p, **kw: LlamaForCausalLM.from_pretrained(p, torch_dtype=torch.float32))
        AD.ValleyLlamaForCausalLM = types.SimpleNamespace(from_pretrained=lambda p, **kw: vm.ValleyLlamaForCausalLM.from_pretrained(p, torch_dtype=torch.float32))
        AD.apply_delta(bdir, tdir, ddir)
        target = vm.ValleyLlamaForCausalLM.from_pretrained(tdir, torch_dtype=torch.float32).state_dict()
    keep = {k: v.numpy() for k, v in target.items() if "rotary" not in k}
    np.savez_compressed(os.path.join(GOLD, "g8_apply_delta.npz"), **keep)
    print("g8_apply_delta:", len(keep), "tensors")


def gen_conversation():
    saved = list(sys.path)
    sys.path[:] = ["/root/reference/valley"] + saved
    try:
        for k in [k for k in sys.modules if k == "conversation"]:
            del sys.modules[k]
        import conversation as RC
    finally:
        sys.path[:] = saved
    assert RC.__file__.startswith("/root/reference/"), RC.__file__
    out = {}
    for name in ("v1", "multimodal_video"):
        conv = RC.conv_templates[name].copy()
        steps = [conv.get_prompt()]
        conv.append_message(conv.roles[0], "what happens?\n<im_start><im_patch><im_end>")
        steps.append(conv.get_prompt())
        conv.append_message(conv.roles[1], None)
        steps.append(conv.get_prompt())
        conv.messages[-1][1] = "a dog runs\n"
        conv.append_message(conv.roles[0], ("and then?", "/tmp/x.mp4", "Crop"))
        steps.append(conv.get_prompt())
        out[name] = {"prompts": steps, "sep": conv.sep, "roles": list(conv.roles), "offset": conv.offset}
    json.dump(out, open(os.path.join(GOLD, "g9_conversation.json"), "w"), indent=0)
    print("g9_conversation:", {k: len(v["prompts"][-1]) for k, v in out.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    gen_ops()
    gen_conversation()
    vm = import_reference()
    gen_decode2(vm)
    gen_apply_delta(vm)
