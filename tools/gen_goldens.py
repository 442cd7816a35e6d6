#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE itself (authoring container only).

Imports /root/reference's ``valley.model.valley_model`` (with empty stub modules for the I/O
dependencies this image lacks: torchvision, decord, cv2, skimage — they are used only by
``load_video``, never by the math) on top of the installed ``transformers``, builds tiny
random-weight models from ``valley_amd.weights``' deterministic tensors, runs the reference
``forward`` and writes inputs' *descriptions* and outputs to ``tests/golden/*.npz``.

The fixtures hold data only (ids, masks, outputs).  Weights and pixels are regenerated from
(seed, name, shape) by the tests.  Nothing here runs on the GPU box; /root/reference does not
exist there.

Usage:  python tools/gen_goldens.py            (rewrites tests/golden/)
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

import numpy as np
import torch
import transformers  # noqa: F401  (must be imported before the stubs)
from transformers import CLIPVisionConfig, CLIPVisionModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from valley_amd import weights as W  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# ---- the tiny golden configuration (shared with tests/golden_cfg.py) -------------------------
from tests.golden_cfg import (GCFG, golden_ids, golden_pixels, llama_state, vision_state,  # noqa: E402
                              extra_pool_state)


def import_reference():
    for name, attrs in [("torchvision", []), ("torchvision.transforms",
                                               ["Compose", "ColorJitter", "RandomApply", "RandomGrayscale", "Resize"]),
                        ("decord", []), ("cv2", []), ("skimage", []), ("skimage.transform", [])]:
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        for a in attrs:
            setattr(m, a, object)
        sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    # This repo ships its own drop-in ``valley`` package (a regular package, which would shadow the
    # reference's namespace package whatever the path order): take the repo root (and cwd) off sys.path
    # while the reference is imported, and make sure the module that comes back IS the reference's file.
    for k in [k for k in sys.modules if k == "valley" or k.startswith("valley.")]:
        del sys.modules[k]
    saved = list(sys.path)
    sys.path[:] = ["/root/reference"] + [p for p in saved if os.path.abspath(p or ".") != ROOT]
    try:
        import valley.model.valley_model as vm
        import valley.data.video_transform  # noqa: F401  (used by tools/gen_preprocess_goldens.py)
    finally:
        sys.path[:] = saved
    assert vm.__file__.startswith("/root/reference/"), vm.__file__
    return vm


def build_reference(vm, method: str):
    c = GCFG
    cfg = vm.ValleyConfig(vocab_size=c["vocab"], hidden_size=c["H"], intermediate_size=c["I"],
                          num_hidden_layers=c["L"], num_attention_heads=c["heads"],
                          num_key_value_heads=c["heads"], rms_norm_eps=c["eps"],
                          max_position_embeddings=2048, attn_implementation="eager")
    cfg.use_mm_proj = True
    cfg.mm_hidden_size = 1024
    cfg.mm_vision_select_layer = -2
    if method == "temporal_importance":
        cfg.use_patch_importance_pooling = True
    if method == "temporal_transformer":
        cfg.use_delta_transformer = True
    model = vm.ValleyLlamaForCausalLM(cfg).eval()
    vcfg = CLIPVisionConfig(hidden_size=1024, num_attention_heads=16, image_size=224, patch_size=14,
                            num_hidden_layers=c["VL"], intermediate_size=c["VI"], attn_implementation="eager")
    tower = CLIPVisionModel(vcfg).eval()
    sd = {k: torch.from_numpy(v) for k, v in vision_state().items()}
    missing, unexpected = tower.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m for m in missing), missing
    for k, v in W.SPECIAL_IDS(c["vocab_text"]).items():
        setattr(tower.config, k, v)
    model.model.vision_tower = tower
    sd = {k: torch.from_numpy(v) for k, v in llama_state().items()}
    sd.update({k: torch.from_numpy(v) for k, v in extra_pool_state(method).items()})
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    missing = [m for m in missing if "vision_tower" not in m and "rotary" not in m]
    assert not missing, missing
    if method == "max":
        model.model.patch_pooling_method = "max"        # only reachable by attribute
    return model


def run(model, ids, images, mask=None, past=None):
    cap = {}
    from transformers import LlamaModel
    orig = LlamaModel.forward

    def spy(self, *a, **k):
        cap["embeds"] = k["inputs_embeds"].detach().clone()
        return orig(self, *a, **k)
    LlamaModel.forward = spy
    try:
        with torch.no_grad():
            out = model(input_ids=ids, images=images, attention_mask=mask, past_key_values=past, use_cache=True)
    finally:
        LlamaModel.forward = orig
    return out, cap.get("embeds")


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    vm = import_reference()
    c = GCFG
    T = c["T"]

    # ---- G1: tower hidden states for 2 frames (all entries) --------------------------------
    model = build_reference(vm, "mean")
    px = torch.from_numpy(golden_pixels(2, "g1"))
    with torch.no_grad():
        hs = model.model.vision_tower(px, output_hidden_states=True).hidden_states
    assert len(hs) == c["VL"] + 1
    np.savez_compressed(os.path.join(GOLD, "g1_tower.npz"),
                        **{f"hs{i}": h.numpy()[:, ::8, ::4].copy() for i, h in enumerate(hs)},
                        hs_sel_full=hs[-2].numpy()[:1].copy())

    # ---- G2/G3/G4: forward with each pooling variant, B=2, sample 1 left-padded ------------
    ids, mask = golden_ids("main")
    images = torch.from_numpy(golden_pixels(2 * T, "main")).view(2, T, 3, 224, 224)
    for method in ("mean", "max", "temporal_importance", "temporal_transformer"):
        model = build_reference(vm, method)
        out, emb = run(model, torch.from_numpy(ids), images, torch.from_numpy(mask))
        np.savez_compressed(os.path.join(GOLD, f"g2_forward_{method}.npz"), ids=ids, mask=mask,
                            embeds=emb.numpy(),
                            logits=out.logits.numpy() if method == "mean" else out.logits.numpy()[:, ::4].copy())
        print(method, "logits", out.logits.shape, float(out.logits.abs().max()))

    model = build_reference(vm, "mean")

    # ---- G3b: text-only sample + multimodal sample, one clip of images ----------------------
    ids, mask = golden_ids("mixed")
    img1 = torch.from_numpy(golden_pixels(T, "mixed")).view(1, T, 3, 224, 224)
    out, emb = run(model, torch.from_numpy(ids), img1, torch.from_numpy(mask))
    np.savez_compressed(os.path.join(GOLD, "g3_mixed.npz"), ids=ids, mask=mask, embeds=emb.numpy(),
                        logits=out.logits.numpy()[:, ::4].copy())

    # ---- G3c: two <im_start> blocks in one sample; frame-count mismatch (silent fallback) ----
    for case in ("two_images", "frame_mismatch"):
        ids, mask = golden_ids(case)
        out, emb = run(model, torch.from_numpy(ids), img1, torch.from_numpy(mask))
        np.savez_compressed(os.path.join(GOLD, f"g3_{case}.npz"), ids=ids, mask=mask, embeds=emb.numpy(),
                            logits=out.logits.numpy()[:, ::4].copy())

    # ---- G3d: error cases ----------------------------------------------------------------------
    errs = {}
    for case in ("cut", "unbalanced"):
        ids, mask = golden_ids(case)
        try:
            run(model, torch.from_numpy(ids), img1, torch.from_numpy(mask))
            errs[case] = "no error"
        except Exception as e:  # noqa: BLE001
            errs[case] = f"{type(e).__name__}: {e}"
        print(case, "->", errs[case])
    np.savez_compressed(os.path.join(GOLD, "g3_errors.npz"), **{k: np.array(v) for k, v in errs.items()})

    # ---- G3e: list-of-clips path with different frame counts ------------------------------------
    ids, mask = golden_ids("list")
    clips = [torch.from_numpy(golden_pixels(2, "list0")), torch.from_numpy(golden_pixels(3, "list1"))]
    out, emb = run(model, torch.from_numpy(ids), clips, torch.from_numpy(mask))
    np.savez_compressed(os.path.join(GOLD, "g3_list.npz"), ids=ids, mask=mask, embeds=emb.numpy(),
                        logits=out.logits.numpy()[:, ::4].copy())

    # ---- G5: prefill + 4 greedy decode steps, manual KV loop (model_worker.py:371-394) ----------
    ids, _ = golden_ids("decode")
    out, _ = run(model, torch.from_numpy(ids), img1)
    past = out.past_key_values
    toks, lasts = [], []
    logits = out.logits
    for _ in range(4):
        last = logits[:, -1, :]
        lasts.append(last.numpy().copy())
        token = last.argmax(-1)
        toks.append(token.numpy().copy())
        ctx = past.get_seq_length()
        with torch.no_grad():
            o = model(input_ids=token[:, None], use_cache=True, attention_mask=torch.ones(1, ctx + 1, dtype=torch.long),
                      past_key_values=past)
        logits, past = o.logits, o.past_key_values
    np.savez_compressed(os.path.join(GOLD, "g5_decode.npz"), ids=ids, prefill_logits=out.logits.numpy(),
                        tokens=np.stack(toks, 1), last_logits=np.stack(lasts, 1))
    print("decode tokens", np.stack(toks, 1))


if __name__ == "__main__":
    main()
