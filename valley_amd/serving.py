"""The decode service loop of the reference worker (valley/serve/model_worker.py:321-426) on the HIP
path: prompt expansion with the clip's real frame count, prefill, per-token KV decode
with greedy / temperature sampling, stop-token and stop-string handling, and ``json\\0`` chunks every
``stream_interval`` tokens.  The HTTP fabric around it (FastAPI worker, controller, gradio) is out of
scope; this generator is what those routes would wrap.

Decode steps run through the hipGraph-captured ``DecodeSession``: the greedy token never leaves the
device between steps; with temperature sampling the step's logits are sampled by torch.multinomial (as in
the reference) and the chosen token is written back into the session's input slot."""
from __future__ import annotations

import json
from typing import Iterator, Optional

import torch

from .decode import DecodeSession
from .valley_model import (DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_VI_END_TOKEN,
                           DEFAULT_VI_START_TOKEN, DEFAULT_VIDEO_FRAME_TOKEN, DEFAULT_VIDEO_TOKEN)


def expand_video_prompt(prompt: str, n_frames: int, use_im_start_end: bool = False) -> str:
    """model_worker.py:338-341."""
    replace_token = DEFAULT_IMAGE_PATCH_TOKEN * 256
    if use_im_start_end:
        replace_token = DEFAULT_IM_START_TOKEN + replace_token + DEFAULT_IM_END_TOKEN + DEFAULT_VI_START_TOKEN + \
            DEFAULT_VIDEO_FRAME_TOKEN * n_frames + DEFAULT_VI_END_TOKEN
    return prompt.replace(DEFAULT_VIDEO_TOKEN, replace_token)


def generate_video_stream(model, tokenizer, params: dict, video: Optional[torch.Tensor] = None, stream_interval: int = 2,
                          context_len: int = 2048, use_graph: bool = True, sampler=None) -> Iterator[bytes]:
    """``params``: prompt, temperature, max_new_tokens, stop (model_worker.py:323-358).  ``video``: preprocessed
    frames [3,T,224,224] (what ``load_video`` returns) or None.  ``sampler(probs) -> token`` replaces the default
    ``torch.multinomial(probs, 1)`` of the temperature branch (:393-394), e.g. to seed it."""
    prompt = params["prompt"]
    ori_prompt = prompt
    images = None
    if video is not None:
        assert prompt.count(DEFAULT_VIDEO_TOKEN) == 1, "Number of video does not match number of <video> tokens in prompt"
        frames = video.permute(1, 0, 2, 3)
        prompt = expand_video_prompt(prompt, frames.shape[0], getattr(model.config, "mm_use_im_start_end", False))
        images = frames.unsqueeze(0)
    temperature = float(params.get("temperature", 1.0))
    max_new_tokens = min(int(params.get("max_new_tokens", 256)), 1024)
    stop_str = params.get("stop", None)
    stop_idx = None
    if stop_str is not None:
        stop_idx = tokenizer(stop_str).input_ids
        stop_idx = stop_idx[0] if len(stop_idx) == 1 else None
    input_ids = tokenizer(prompt).input_ids
    max_src_len = context_len - max_new_tokens - 8
    input_ids = input_ids[-max_src_len:]
    pred_ids = []
    dev = model.device
    ll = model.get_model().llama
    cache = ll.new_cache(1, min(context_len, len(input_ids) + max_new_tokens + 1))
    sess = None
    ret = None
    for i in range(max_new_tokens):
        if i == 0:
            out = model(input_ids=torch.as_tensor([input_ids], device=dev), use_cache=True, images=images, past_key_values=cache)
            last = out.logits[0, -1]
        else:
            if sess is None:
                sess = DecodeSession(ll, cache, use_graph=use_graph)
                sess.begin(torch.as_tensor([token], device=dev))
            else:
                sess.tok.copy_(torch.as_tensor([token], device=dev, dtype=torch.int32))
            sess.step()
            last = sess.logits[0, :ll.V]
        if temperature < 1e-4:
            token = int(torch.argmax(last))
        else:
            probs = torch.softmax(last / temperature, dim=-1)
            token = int(torch.multinomial(probs, num_samples=1)) if sampler is None else int(sampler(probs))
        pred_ids.append(token)
        if stop_idx is not None and token == stop_idx:
            stopped = True
        elif token == getattr(tokenizer, "eos_token_id", None):
            stopped = True
        else:
            stopped = False
        if i % stream_interval == 0 or i == max_new_tokens - 1 or stopped:
            cur_out = tokenizer.decode(pred_ids, skip_special_tokens=True)
            pos = cur_out.rfind(stop_str) if stop_str is not None else -1
            if pos != -1:
                cur_out = cur_out[:pos]
                stopped = True
            ret = {"text": ori_prompt + cur_out, "error_code": 0}
            yield json.dumps(ret).encode() + b"\0"
        if stopped or cache.seq_len + 1 > cache.ctx_max:
            break
    if sess is not None:
        sess.check()                                         # the decode launches' ticket counters / abort word, once per request


class ContinuousBatcher:
    """Continuous batching over ONE hipGraph-captured decode step: up to ``slots`` (<= 8) requests
    share a KV cache [slots, heads, ctx_max, 128] and a captured step of batch ``slots``; a request joins at any step
    (its prompt is prefilled into its slot's cache rows by the ordinary MFMA prefill), leaves at any step, and the
    weight stream of every decode step — the whole cost of a step at these batch sizes (HBM-bound GEMV) — is shared by
    all live requests.  The reference's worker serialises requests behind a semaphore and runs the loop of
    model_worker.py:371-394 once per request; the per-request token sequence here is the same as that loop's.

    Every slot has its own position (device int32, advanced by the captured step) and its own key-validity row, so the
    slots are fully independent sequences: ``vly_decode_attention_rows``.  Idle slots run along (their rows are computed
    and ignored: the GEMV cost does not depend on the row count), clamped inside their cache rows."""

    def __init__(self, model, slots: int = 4, ctx_max: int = 1024, use_graph: bool = True):
        if not 1 <= slots <= 8:
            raise ValueError("1 <= slots <= 8 (the decode step streams weights with the GEMV kernels)")
        self.model, self.ll = model, model.get_model().llama
        if ctx_max > self.ll.max_positions:
            # positions index the RoPE tables, which hold max_position_embeddings rows
            raise ValueError(f"ctx_max {ctx_max} exceeds the model's {self.ll.max_positions} positions")
        self.slots, self.ctx_max = slots, ctx_max
        self.full = []                                           # slots released by step() because their cache rows filled up
        self.cache = self.ll.new_cache(slots, ctx_max)
        self.cache.key_valid = torch.ones((slots, ctx_max), dtype=torch.uint8, device=self.ll.device)
        self.sess = DecodeSession(self.ll, self.cache, use_graph=use_graph, per_row_positions=True)
        self.live = [False] * slots
        self.length = [0] * slots                                # tokens in each slot's cache (host mirror of sess.pos)
        self._captured = False

    def free_slots(self):
        return [i for i, v in enumerate(self.live) if not v]

    def add(self, input_ids, images=None, attention_mask=None, first_token: Optional[int] = None) -> int:
        """Prefill one request (input_ids [1, S]) into a free slot; returns the slot.  The first generated token is the
        prefill's argmax unless ``first_token`` is given (a caller that samples)."""
        free = self.free_slots()
        if not free:
            raise RuntimeError("no free slot")
        slot = free[0]
        ids = torch.as_tensor(input_ids, device=self.ll.device).view(1, -1)
        S = ids.shape[1]
        if S + 1 > self.ctx_max:
            raise ValueError("prompt does not fit the slot")
        self.cache.key_valid[slot] = 1
        row = type(self.cache).rows_of(self.cache, slot, slot + 1)
        out = self.model(input_ids=ids, images=images, attention_mask=attention_mask, past_key_values=row, use_cache=True)
        tok = int(out.logits[0, -1].argmax()) if first_token is None else int(first_token)
        self.sess.pos[slot:slot + 1].fill_(S)
        self.sess.tok[slot:slot + 1].fill_(tok)
        self.live[slot], self.length[slot] = True, S
        self.last_prefill_logits = out.logits[0, -1]
        return slot

    def step(self) -> dict:
        """One decode step for every live slot: feeds each slot's current token, returns {slot: next greedy token}.
        ``self.sess.logits[slot, :V]`` holds that slot's logits (a sampling caller overwrites ``self.sess.tok[slot]``)."""
        if not self._captured:
            self.sess.begin()
            self._captured = True
        # a slot whose cache rows are full leaves the batch here (reported in self.full) instead of failing the shared
        # step of every other live request
        self.full = [i for i in range(self.slots) if self.live[i] and self.length[i] + 1 > self.ctx_max]
        for i in self.full:
            self.release(i)
        if not any(self.live):
            return {}
        self.sess.step()
        toks = self.sess.tok.tolist()                            # one D2H read per step for all requests
        out = {}
        for i in range(self.slots):
            if self.live[i]:
                self.length[i] += 1
                out[i] = toks[i]
        return out

    def release(self, slot: int) -> None:
        if self._captured and self.live[slot]:
            self.sess.check()                                    # once per leaving request (the step's D2H read has synchronised already)
        self.live[slot] = False
