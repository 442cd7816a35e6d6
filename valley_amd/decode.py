"""hipGraph-captured autoregressive decode step (BASELINE.json configs[4]; the loop of
valley/serve/model_worker.py:380-394 and of HF ``generate`` behind valley_model.py:432).

One decode step = embedding gather of the current token -> L x [RMSNorm + q|k|v GEMV, RoPE + KV append +
decode attention (every head split over four workgroups at batch <= 2), (merge +) o GEMV(+res), RMSNorm + gate/up GEMV with SwiGLU, down GEMV(+res)] -> RMSNorm +
lm_head GEMV -> argmax -> position += 1.  Every kernel is HBM-bound weight/KV streaming, 5 launches
per layer (the norms ride inside the GEMV that consumes them at batch <= 2; 7 otherwise): launched eagerly from Python the step would be host-bound (>300 launches x ~15 us), so the
step is captured ONCE into a hipGraph and replayed.  Static shapes are what capture needs: the KV
cache is pre-allocated to ctx_max, and the only thing that changes between replays — the position —
lives on the device (``pos``) and is read by vly_rope_kv / vly_llama_attention through their
``past_len_dev`` argument."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import ops
from . import runtime
from .llama import HipKVCache, HipLlama


FUSE_NORM = os.environ.get("VALLEY_DECODE_FUSE_NORM", "1") != "0"
SPLIT_ATTN = os.environ.get("VALLEY_DECODE_SPLIT_ATTN", "1") != "0"       # flash-decoding split + merge inside the o GEMV (B <= 2)
# round 4: every decoder layer of the step in ONE persistent launch (vly_decode_layers: grid barriers between the five phases of a
# layer, the next phase's first weight units requested before the barrier; BIT-identical to the five launches per layer,
# tests/test_decode_persistent_gpu.py).  Measured (13B, 256 tokens, same box, profiles/r04): 198.7 tokens/s against the launches'
# 208-209 — its weight loops stream at 6.9-7 TB/s, but a phase boundary inside the launch (arrival skew 3.5-8 us + barrier 1.5-2 us +
# activation hand-off and norm 1.4-2.8 us) costs no less than a kernel boundary (1.3 us + ramp), and requests issued before the
# barrier land before it ends (DESIGN.md "decode: the persistent step").  Kept as an option: VALLEY_DECODE_PERSISTENT=1.
PERSISTENT = os.environ.get("VALLEY_DECODE_PERSISTENT", "0") != "0"
# round 4: where the split attention's partials are merged — "attn": by the last workgroup of a head inside the attention launch
# (vly_decode_attention_merged; the o projection is then a plain GEMV), "oproj": in the o GEMV's prologue (round 3).  Same bits.
# ("oproj" and the persistent step need the experimental library: VALLEY_EXPERIMENTAL=1, include/valley_hip.h's EXPERIMENTAL prototypes)
MERGE_IN = os.environ.get("VALLEY_DECODE_MERGE", "attn")
SPLIT_ROWS = os.environ.get("VALLEY_DECODE_SPLIT_ROWS", "1") != "0"       # round 5: the merged split attention for 3 .. 8 rows as well

class DecodeSession:
    def __init__(self, llama: HipLlama, cache: HipKVCache, use_graph: bool = True, per_row_positions: bool = False):
        """``per_row_positions``: every batch row is an independent sequence at its own position (``pos`` is int32 [B]
        and advances by one per step for every row) — the captured step of valley_amd.serving.ContinuousBatcher."""
        self.ll, self.cache = llama, cache
        B, d = cache.batch, llama.device
        if B > 8:
            raise ValueError("decode sessions stream weights with the GEMV kernel: batch <= 8")
        self.B = B
        self.per_row = per_row_positions
        self.tok = torch.zeros((B,), dtype=torch.int32, device=d)          # token fed to the next step
        self.pos = torch.zeros((B if per_row_positions else 1,), dtype=torch.int32, device=d)   # on the device: replays need no patching
        self.h = torch.empty((B, llama.H), dtype=torch.float32, device=d)
        bf = runtime.HALF
        self.x = torch.empty((B, llama.H), dtype=bf, device=d)
        self.qkv = torch.empty((B, 3 * llama.H), dtype=bf, device=d)
        self.att = torch.empty((B, llama.H), dtype=bf, device=d)
        self.partials = ops.decode_partials(B, llama.heads, d)
        self.arrivals = torch.zeros((B * llama.heads,), dtype=torch.int32, device=d)     # tickets of vly_decode_attention_merged
        self.mlp = torch.empty((B, llama.I), dtype=bf, device=d)
        self.logits = torch.empty((B, llama.Vpad), dtype=torch.float32, device=d)
        # the persistent form takes the whole GPU (one workgroup per CU, all resident): shapes it supports, and only with the
        # fused norm / split attention arithmetic it reproduces
        self.persistent = (PERSISTENT and FUSE_NORM and SPLIT_ATTN and llama.heads * 128 == llama.H
                           and ops.decode_layers_ok(B, llama.H, llama.heads, llama.I))
        if self.persistent:
            self.mlp32 = torch.empty((B, llama.I), dtype=torch.float32, device=d)
            self.sync = torch.zeros((ops.DECODE_SYNC_WORDS,), dtype=torch.int32, device=d)
            self.table = None
            self._table_gen = None
        self.use_graph = use_graph
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._gen = cache.generation

    def _enqueue_step(self):
        ll, c = self.ll, self.cache
        B = self.B
        # the three norm -> projection seams as one launch each where the fused kernel takes the shape (bit-identical either way;
        # VALLEY_DECODE_FUSE_NORM=0 keeps the pairs, for A/B runs)
        fused = FUSE_NORM and ops.gemv_rmsnorm_ok(B, ll.H)
        # every head over four workgroups.  Merged inside the attention launch (the default) it does not depend on the o GEMV's form, so
        # three to eight rows take it too: at eight requests decode_fused_kernel's 320 workgroups of 512 threads are 1.25 rounds of one
        # workgroup per CU (23 us per layer, 15 % of the step); 1280 quarter-head workgroups stream the same K / V evenly
        # (VALLEY_DECODE_SPLIT_ROWS=0: the one-workgroup-per-head kernel for more than two rows, A/B runs)
        split = SPLIT_ATTN and ll.heads * 128 == ll.H and (ops.gemv_rmsnorm_ok(B, ll.H) or (SPLIT_ROWS and MERGE_IN == "attn"))
        ops.embed_splice(self.tok, ll.embed, None, out=self.h)
        if self.persistent:
            if self.table is None or self._table_gen != c.generation:      # raw pointers: the cache's storage may have moved
                self.table = ops.decode_layer_table(ll.layers, c.k, c.v, ll.device)
                self._table_gen = c.generation
            ops.decode_layers(self.table, self.h, self.qkv, self.partials, self.mlp32, ll.cos, ll.sin, c.key_valid, self.pos,
                              self.per_row, ll.heads, ll.I, ll.eps, c.ctx_max, self.sync)
        for li in range(0 if self.persistent else ll.L):
            L = ll.layers[li]
            if fused:
                ops.gemv_rmsnorm(self.h, L["ln1"], ll.eps, L["w_qkv"], out=self.qkv)      # input_layernorm inside the q|k|v GEMV
            else:
                ops.rmsnorm(self.h, L["ln1"], ll.eps, out=self.x)
                ops.gemv(self.x, L["w_qkv"], out=self.qkv)
            if split:                                            # every head over four workgroups; the o GEMV merges
                if MERGE_IN == "attn":
                    ops.decode_attention_split(self.qkv, c.k[li], c.v[li], ll.cos, ll.sin, c.key_valid, B, ll.heads, 0, self.partials,
                                               past_dev=self.pos, per_row=self.per_row, out=self.att, arrivals=self.arrivals)
                    ops.gemv(self.att, L["w_o"], residual=self.h, out=self.h)
                else:
                    ops.decode_attention_split(self.qkv, c.k[li], c.v[li], ll.cos, ll.sin, c.key_valid, B, ll.heads, 0,
                                               self.partials, past_dev=self.pos, per_row=self.per_row)
                    ops.gemv_attnmerge(self.partials, L["w_o"], residual=self.h, out=self.h)
            elif self.per_row:
                ops.decode_attention_rows(self.qkv, c.k[li], c.v[li], ll.cos, ll.sin, c.key_valid, B, ll.heads, self.pos, out=self.att)
            else:
                ops.decode_attention(self.qkv, c.k[li], c.v[li], ll.cos, ll.sin, c.key_valid, B, ll.heads, 0, out=self.att,
                                     past_dev=self.pos)          # RoPE + KV append + attention in one launch
            if not split:
                ops.gemv(self.att, L["w_o"], residual=self.h, out=self.h)
            if fused:
                ops.gemv_rmsnorm(self.h, L["ln2"], ll.eps, L["w_gu"], epilogue=ops.EPI_SWIGLU, out=self.mlp)
            else:
                ops.rmsnorm(self.h, L["ln2"], ll.eps, out=self.x)
                ops.gemv(self.x, L["w_gu"], epilogue=ops.EPI_SWIGLU, out=self.mlp)
            ops.gemv(self.mlp, L["w_down"], residual=self.h, out=self.h)
        if fused:
            ops.gemv_rmsnorm(self.h, ll.norm, ll.eps, ll.lm_head, out=self.logits)
        else:
            ops.rmsnorm(self.h, ll.norm, ll.eps, out=self.x)
            ops.gemv(self.x, ll.lm_head, out=self.logits)
        # greedy next token straight into the input slot of the next step (the V-padding columns of the
        # lm_head buffer are excluded through the row stride)
        ops.argmax(self.logits[:, :ll.V], out=self.tok)
        ops.incr_i32(self.pos, 1)

    def begin(self, first_token: Optional[torch.Tensor] = None):
        """Call after the prefill filled ``cache``: sets the device position and the first input token (per-row sessions
        manage ``pos`` / ``tok`` per slot themselves and call this once, to capture)."""
        if not self.per_row:
            self.pos.fill_(self.cache.seq_len)
            self.tok.copy_(first_token.to(torch.int32).view(-1))
            if self.cache.key_valid is not None:
                self.cache.key_valid[:, self.cache.seq_len:] = 1       # generated positions are always attended
        if self.use_graph and (self.graph is None or self._gen != self.cache.generation):
            self._capture()

    def _capture(self):
        """Warm-up outside capture on a side stream (module loading, lazy init), then capture ONE step; the device-side
        position / token are restored, so capturing is invisible to the sequence.  Re-run whenever the cache's storage
        moved (HipKVCache.reserve grew it): the graph holds raw pointers."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        pos0, tok0 = self.pos.clone(), self.tok.clone()
        with torch.cuda.stream(s):
            self._enqueue_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.pos.copy_(pos0)
        self.tok.copy_(tok0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._enqueue_step()
        self.graph = g
        self._gen = self.cache.generation

    def check(self) -> None:
        """Raise if a workgroup of the persistent launch gave up at a grid barrier in the last step (it needs every CU: a kernel
        of another stream was holding some), or if the merged attention's ticket counters are not back at zero (a launch that
        did not complete).  Costs a device-to-host copy: callers check once per generation, tests per step."""
        if int(self.arrivals.abs().sum().item()) != 0:
            self.arrivals.zero_()
            raise RuntimeError("vly_decode_attention_merged: ticket counters not at zero after a step (an attention launch did "
                               "not complete); the step's output is invalid")
        if self.persistent and int(self.sync[ops.DECODE_SYNC_ABORT].item()) != 0:
            self.sync.zero_()                                    # the barrier counters are inconsistent after an abort
            raise RuntimeError("vly_decode_layers: a grid barrier timed out (not every workgroup was resident); the step's "
                               "output is invalid — rerun with VALLEY_DECODE_PERSISTENT=0 or keep the GPU to this stream")

    def step(self) -> torch.Tensor:
        """Run one decode step; returns the (device) int32 [B] buffer holding the newly chosen token.
        ``self.logits[:, :V]`` holds that step's logits (for temperature sampling on the host side)."""
        if not self.per_row:
            try:
                self.cache.reserve(self.cache.seq_len + 1)           # grows a model-sized cache (new storage -> new graph)
            except ValueError:
                raise ValueError("KV cache full") from None
            if self.cache.key_valid is not None and self.cache.key_valid.shape[1] != self.cache.ctx_max:
                raise RuntimeError("key_valid out of step with the cache")
            if self.graph is not None and self._gen != self.cache.generation:
                self._capture()
        if self.graph is not None:
            self.graph.replay()
        else:
            self._enqueue_step()
        self.cache.seq_len += 1
        return self.tok
