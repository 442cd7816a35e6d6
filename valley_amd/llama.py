"""Llama decoder (prefill + KV-cache decode) on the HIP kernels.

Replaces HF ``LlamaModel.forward`` as called by ``ValleyLlamaModel.forward``
(valley/model/valley_model.py:249-254 -> hf:llama/modeling_llama.py:347-417) and the ``lm_head``
(valley_model.py:264,304-305).  Per layer: RMSNorm -> fused q|k|v GEMM -> RoPE + KV append ->
causal attention -> o_proj GEMM (+residual) -> RMSNorm -> fused gate/up GEMM with SwiGLU epilogue
-> down_proj GEMM (+residual).  MHA only (7B/13B have no GQA), head_dim 128.

HBM layout: residual stream fp32 [B*S, H]; weights bf16, q/k/v concatenated to [3H, H], gate/up
row-interleaved to [2I, H] so that the SwiGLU pair sits in one lane; KV cache bf16
[B, heads, ctx_max, 128] per layer (the legacy HF tuple layout, so ``past[l][0]`` is a plain view).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops, runtime


def _dev(t, device, dtype):
    if not isinstance(t, torch.Tensor):
        t = torch.from_numpy(np.ascontiguousarray(t))
    return t.to(device=device, dtype=dtype).contiguous()


class HipKVCache:
    """Pre-allocated KV cache.  Supports the legacy indexing the serving loop uses
    (``past_key_values[0][0].shape[-2]``, serve/model_worker.py:381) and ``get_seq_length()``."""

    def __init__(self, layers: int, batch: int, heads: int, ctx_max: int, device):
        runtime.bind_half("HipKVCache")                      # 16-bit tensors exist from here on: the storage type is final
        self.k = [torch.zeros((batch, heads, ctx_max, 128), dtype=runtime.HALF, device=device) for _ in range(layers)]
        self.v = [torch.zeros((batch, heads, ctx_max, 128), dtype=runtime.HALF, device=device) for _ in range(layers)]
        self.seq_len = 0
        self.ctx_max = ctx_max
        self.batch = batch
        self.key_valid: Optional[torch.Tensor] = None      # uint8 [B, ctx_max] when a padding mask was given
        self.generation = 0                                  # bumped when the storage moves (a captured decode graph must be rebuilt)
        self.growable = False                                # set by the model for caches IT sized (the caller never said how deep)
        self.limit = ctx_max

    def reserve(self, n: int) -> None:
        """Make room for ``n`` positions.  A cache the model created on the caller's behalf (use_cache=True without
        past_key_values, the reference's worker loop, serve/model_worker.py:373-387) starts at prompt length + 256 instead
        of max_position_embeddings (13B: 1.7 GB per sequence at 2048) and doubles when it fills up — the HF DynamicCache the
        reference uses grows the same way; a cache the caller sized (new_cache(B, ctx)) never moves."""
        if n <= self.ctx_max:
            return
        if not self.growable or n > self.limit:
            raise ValueError(f"KV cache overflow: {n} > {self.ctx_max}")
        new_ctx = min(self.limit, max(n, 2 * self.ctx_max))
        for buf in (self.k, self.v):
            for i, t in enumerate(buf):
                g = torch.zeros((t.shape[0], t.shape[1], new_ctx, 128), dtype=t.dtype, device=t.device)
                g[:, :, :self.seq_len] = t[:, :, :self.seq_len]
                buf[i] = g
        if self.key_valid is not None:
            kv = torch.ones((self.batch, new_ctx), dtype=torch.uint8, device=self.key_valid.device)
            kv[:, :self.ctx_max] = self.key_valid
            self.key_valid = kv
        self.ctx_max = new_ctx
        self.generation += 1

    @classmethod
    def rows_of(cls, parent: "HipKVCache", b0: int, b1: int) -> "HipKVCache":
        """A cache object over batch rows [b0, b1) of ``parent`` (shared storage: the batch is the outermost dimension,
        so the slices are contiguous) — how a request is prefilled into its slot of a continuous-batching cache."""
        c = cls.__new__(cls)
        c.k = [t[b0:b1] for t in parent.k]
        c.v = [t[b0:b1] for t in parent.v]
        c.seq_len, c.ctx_max, c.batch = 0, parent.ctx_max, b1 - b0
        c.key_valid = None if parent.key_valid is None else parent.key_valid[b0:b1]
        c.generation, c.growable, c.limit = 0, False, parent.ctx_max
        return c

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.seq_len

    def __len__(self):
        return len(self.k)

    def __bool__(self):                                      # an allocated cache is "truthy" only once filled
        return self.seq_len > 0

    def __getitem__(self, layer: int):
        return (self.k[layer][:, :, :self.seq_len], self.v[layer][:, :, :self.seq_len])

    def __iter__(self):
        for i in range(len(self.k)):
            yield self[i]


class HipLlama:
    def __init__(self, hidden: int, heads: int, intermediate: int, layers: int, vocab: int, eps: float,
                 rope_theta: float = 10000.0, max_positions: int = 2048, device="cuda:0",
                 pack_weights: Optional[bool] = None):
        if hidden != heads * 128:
            raise ValueError("HIP Llama path requires head_dim == 128 (hidden = heads*128)")
        if hidden % 64 or intermediate % 64:
            raise ValueError("hidden and intermediate sizes must be multiples of 64")
        runtime.bind_half("HipLlama")                        # its weights are allocated in runtime.HALF: final from here on
        self.H, self.heads, self.I, self.L, self.V, self.eps = hidden, heads, intermediate, layers, vocab, eps
        self.Vpad = (vocab + 7) // 8 * 8
        self.device = torch.device(device)
        self.max_positions = max_positions
        # hf:llama/modeling_llama.py:95-124 : inv_freq = theta^(-2i/d), angle = pos * inv_freq, fp32
        inv = 1.0 / (rope_theta ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))
        ang = torch.arange(max_positions, dtype=torch.float32)[:, None] * inv[None]
        self.cos, self.sin = ang.cos().contiguous().to(self.device), ang.sin().contiguous().to(self.device)
        self.layers: List[Dict[str, torch.Tensor]] = []
        self.loaded = False
        self._ws = {}
        # prefill reads the four projections of every layer from a second, block-ordered copy (ops.PackedWeight):
        # contiguous weight tiles in HBM for the MFMA kernels (same box, c2: prefill 19.2 -> 18.8 ms; down-proj +5 %,
        # gate/up +2.5 %); decode keeps streaming the row-major copy.  Costs the projection bytes a second time
        # (13 GB at 7B, 26 GB at 13B of the 288 GB); VALLEY_PACK_WEIGHTS=0 or pack_weights=False keeps one copy.
        self.pack_weights = os.environ.get("VALLEY_PACK_WEIGHTS", "1") == "1" if pack_weights is None else bool(pack_weights)
        self.packed: List[Dict[str, "ops.PackedWeight"]] = []

    def _pack(self):
        self.packed = [{k: ops.PackedWeight(L[k]) for k in ("w_qkv", "w_o", "w_gu", "w_down")} for L in self.layers] \
            if self.pack_weights else []

    # ---- weights -------------------------------------------------------------------------------
    @staticmethod
    def _interleave(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
        I, H = gate.shape
        return torch.stack([gate, up], dim=1).reshape(2 * I, H).contiguous()

    def load_state_dict(self, sd: Dict) -> "HipLlama":
        """Reference key names (SURVEY.md §5 weight-loading contract)."""
        d, bf, f32 = self.device, runtime.HALF, torch.float32
        self.embed = _dev(sd["model.embed_tokens.weight"], d, bf)
        self.layers = []
        for i in range(self.L):
            p = f"model.layers.{i}."
            L = {}
            L["ln1"] = _dev(sd[p + "input_layernorm.weight"], d, f32)
            L["ln2"] = _dev(sd[p + "post_attention_layernorm.weight"], d, f32)
            L["w_qkv"] = torch.cat([_dev(sd[p + f"self_attn.{n}_proj.weight"], d, bf) for n in "qkv"], 0).contiguous()
            L["w_o"] = _dev(sd[p + "self_attn.o_proj.weight"], d, bf)
            L["w_gu"] = self._interleave(_dev(sd[p + "mlp.gate_proj.weight"], d, bf), _dev(sd[p + "mlp.up_proj.weight"], d, bf))
            L["w_down"] = _dev(sd[p + "mlp.down_proj.weight"], d, bf)
            self.layers.append(L)
        self.norm = _dev(sd["model.norm.weight"], d, f32)
        self.lm_head = torch.zeros((self.Vpad, self.H), dtype=bf, device=d)
        self.lm_head[:self.V] = _dev(sd["lm_head.weight"], d, bf)
        self._pack()
        self.loaded = True
        return self

    def init_random(self, seed: int = 0, std: float = 0.02) -> "HipLlama":
        d, bf, f32 = self.device, runtime.HALF, torch.float32
        g = torch.Generator(device=d).manual_seed(seed)
        rn = lambda shape: (torch.randn(shape, generator=g, device=d, dtype=f32) * std).to(bf)  # noqa: E731
        self.embed = rn((self.V, self.H))
        self.layers = []
        for _ in range(self.L):
            self.layers.append(dict(ln1=torch.ones(self.H, device=d), ln2=torch.ones(self.H, device=d),
                                    w_qkv=rn((3 * self.H, self.H)), w_o=rn((self.H, self.H)),
                                    w_gu=rn((2 * self.I, self.H)), w_down=rn((self.H, self.I))))
        self.norm = torch.ones(self.H, device=d)
        self.lm_head = torch.zeros((self.Vpad, self.H), dtype=bf, device=d)
        self.lm_head[:self.V] = rn((self.V, self.H))
        self._pack()
        self.loaded = True
        return self

    # ---- compute ---------------------------------------------------------------------------------
    def new_cache(self, batch: int, ctx_max: Optional[int] = None) -> HipKVCache:
        return HipKVCache(self.L, batch, self.heads, ctx_max or self.max_positions, self.device)

    def _workspace(self, M: int):
        key = (M, runtime.stream_key())                    # concurrent streams never share activations
        ws = self._ws.get(key)
        if ws is None:
            d, bf = self.device, runtime.HALF
            ws = dict(x=torch.empty((M, self.H), dtype=bf, device=d), qkv=torch.empty((M, 3 * self.H), dtype=bf, device=d),
                      att=torch.empty((M, self.H), dtype=bf, device=d), mlp=torch.empty((M, self.I), dtype=bf, device=d),
                      delta=torch.empty((M, self.H), dtype=bf, device=d), delta2=torch.empty((M, self.H), dtype=bf, device=d))
            if len(self._ws) > 8:
                self._ws.clear()
            self._ws[key] = ws
        return ws

    def forward(self, h: torch.Tensor, B: int, S: int, cache: HipKVCache, n_layers: Optional[int] = None,
                collect: Optional[list] = None, attn: Optional[list] = None) -> torch.Tensor:
        """h fp32 [B*S, H] (modified in place) -> final-norm hidden bf16 [B*S, H].  Appends S
        positions to ``cache``; key validity comes from cache.key_valid.  ``collect`` (a list): HF's ``output_hidden_states``
        (hf:llama/modeling_llama.py LlamaModel.forward) — receives the embeddings, every decoder layer's output (fp32
        copies of the residual stream) and, last, the final-norm output; costs one residual flush + copy per layer.
        ``attn`` (a list): HF's ``output_attentions`` — receives every layer's probabilities, fp32 [B, heads, S, kv_len]
        (ops.attention_probs: a separate pass per layer behind the RoPE launch; the attention kernels themselves are unchanged)."""
        if not self.loaded:
            raise RuntimeError("Llama engine has no weights")
        past = cache.seq_len
        cache.reserve(past + S)                              # grows a model-sized cache, raises for a caller-sized one
        if cache.batch != B:
            raise ValueError("cache batch mismatch")
        ops.sk_check_polled(self.device)                   # a stream-K hand-off failure of an earlier call surfaces here
        with runtime.stream_lock():                        # launch sequences on one stream must not interleave
            out = self._forward_locked(h, B, S, cache, past, n_layers, collect, attn)
            if S > 1:
                ops.sk_poll_async(self.device)
            return out

    def _forward_locked(self, h, B, S, cache, past, n_layers, collect=None, attn=None):
        M = B * S
        ws = self._workspace(M)
        kv = cache.key_valid                                # uint8 [B, ctx_max] or None (row stride = ctx_max)
        nl = self.L if n_layers is None else n_layers
        fused = M > 8            # prefill: residual adds ride on the norm kernels; decode keeps the GEMV epilogue
        d2 = None                  # second split-K partial of the pending sub-layer output (ops.gemm2), if any
        pending = False            # ws["delta"] (+ d2) holds a sub-layer output that has not been added to h yet
        if collect is not None:
            collect.append(h.clone())                       # hidden_states[0]: the (spliced) input embeddings
        for li in range(nl):
            L = self.layers[li]
            W = self.packed[li] if (fused and self.packed) else L      # projection weights as this pass reads them
            if pending:
                ops.add_norm(h, ws["delta"], L["ln1"], None, self.eps, out=ws["x"], rms=True, delta2=d2)
            else:
                ops.rmsnorm(h, L["ln1"], self.eps, out=ws["x"])
            if attn is not None:                            # output_attentions: the unfused sequence leaves the rotated q in qkv
                ops.gemm(ws["x"], W["w_qkv"], out=ws["qkv"])
                ops.rope_kv(ws["qkv"], cache.k[li], cache.v[li], self.cos, self.sin, B, S, self.heads, past)
                ops.llama_attention(ws["qkv"], cache.k[li], cache.v[li], kv, B, S, self.heads, past, out=ws["att"])
                attn.append(ops.attention_probs(ws["qkv"], cache.k[li], kv, B, S, self.heads, past))
            elif S == 1:                                    # one-token step: RoPE + append + attention fused
                ops.gemm(ws["x"], W["w_qkv"], out=ws["qkv"])
                ops.decode_attention(ws["qkv"], cache.k[li], cache.v[li], self.cos, self.sin, kv, B, self.heads, past,
                                     out=ws["att"])
            else:
                # prefill: q|k|v GEMM, then RoPE + KV append (VALLEY_FUSE_ROPE, default auto: both in the GEMM's epilogue where that shape was measured — bit-identical,
                # one launch less; +0.3 % on c3)
                ops.gemm_qkv_rope(ws["x"], W["w_qkv"], ws["qkv"], ops.RopeKV(cache.k[li], cache.v[li], self.cos, self.sin, B, S,
                                                                            self.heads, past))
                ops.llama_attention(ws["qkv"], cache.k[li], cache.v[li], kv, B, S, self.heads, past, out=ws["att"])
            if fused:
                # o_proj / down_proj have fewer output tiles than the chip has CUs at prefill sizes: the tuner may
                # answer with a split-K pair whose two bf16 partials the add+norm kernel sums
                d2 = ws["delta2"] if ops.gemm2(ws["att"], W["w_o"], ws["delta"], ws["delta2"]) == 2 else None
                ops.add_norm(h, ws["delta"], L["ln2"], None, self.eps, out=ws["x"], rms=True, delta2=d2)
            else:
                ops.gemm(ws["att"], L["w_o"], residual=h, out=h)
                ops.rmsnorm(h, L["ln2"], self.eps, out=ws["x"])
            ops.gemm(ws["x"], W["w_gu"], epilogue=ops.EPI_SWIGLU, out=ws["mlp"])
            if fused:
                d2 = ws["delta2"] if ops.gemm2(ws["mlp"], W["w_down"], ws["delta"], ws["delta2"]) == 2 else None
                pending = True
            else:
                ops.gemm(ws["mlp"], L["w_down"], residual=h, out=h)
            if collect is not None and li + 1 < nl:
                if pending:                                 # flush: h is a complete hidden state again
                    ops.add_norm(h, ws["delta"], None, None, self.eps, rms=True, delta2=d2)
                    pending = False
                collect.append(h.clone())
        cache.seq_len = past + S
        if pending:
            out = ops.add_norm(h, ws["delta"], self.norm, None, self.eps, out=ws["x"], rms=True, delta2=d2)
        else:
            out = ops.rmsnorm(h, self.norm, self.eps, out=ws["x"])
        if collect is not None:
            collect.append(out.float())                     # HF: the last entry is taken AFTER the final norm
        return out

    def logits(self, x: torch.Tensor) -> torch.Tensor:
        """x bf16 [M,H] -> fp32 [M,V] (a view of the V-padded GEMM output)."""
        out = ops.gemm(x, self.lm_head, out_dtype=torch.float32)
        return out[:, :self.V]
