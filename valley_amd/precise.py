"""fp32 "precise" engines (VALLEY_PRECISION=fp32 / ``ValleyConfig.valley_precision = "fp32"``).

Same operators, same call structure and same reference lines as valley_amd/vision_tower.py and valley_amd/llama.py, but
every tensor — weights, activations, residual stream, KV cache — is fp32 and every contraction runs on the exact
f32-input MFMA (valley_amd/csrc/precise_f32.hip) — or, with VALLEY_F32_GEMM=x3 (valley_amd/ops_f32.py, round 6), the GEMMs run as three
16-bit partial products (hi.hi + hi.lo + lo.hi, fp32 accumulation) on the production MFMA kernels: logits 5e-5 from the reference's on
the fixtures at ~4x the rate of the exact mode (attention, norms, RoPE and softmax stay on the fp32 kernels either way).  This is the mode in which BASELINE.json's "logits within 1e-3 of
reference" is demonstrated against the fp32 reference fixtures (tests/test_precise_gpu.py); the bf16 engines are the
production path and are held to the same fixtures at their own stated tolerance and to the bf16-rounded oracle.
~1/16 of the bf16 MFMA rate by construction, so it is meant for validation-sized runs, not for the benchmark."""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops, ops_f32 as F, runtime
from .vision_tower import VisionConfig

F32 = torch.float32


def _dev(t, device):
    if not isinstance(t, torch.Tensor):
        t = torch.from_numpy(np.ascontiguousarray(t))
    return t.to(device=device, dtype=F32).contiguous()


class PreciseCLIPVisionTower:
    """hf:clip/modeling_clip.py:138-218, 259-383, 594-656 to ``hidden_states[select_layer]`` in fp32."""
    precision = "fp32"

    def __init__(self, config: Optional[VisionConfig] = None, device="cuda:0"):
        self.config = config or VisionConfig()
        c = self.config
        if c.hidden_size != 1024 or c.num_attention_heads != 16 or c.image_size != 224 or c.patch_size != 14:
            raise ValueError("the HIP tower supports the ViT-L/14 geometry only (1024 wide, 16 heads, 224/14)")
        if c.hidden_act != "quick_gelu":
            raise ValueError("only quick_gelu towers are supported")
        self.device = torch.device(device)
        self.layers: List[Dict[str, torch.Tensor]] = []
        self.loaded = False

    def load_state_dict(self, sd: Dict, prefix: str = "", truncated_ok: bool = False, shallow_ok: bool = False):
        if prefix == "" and any(k.startswith("vision_model.") for k in sd):
            prefix = "vision_model."
        g = lambda k: _dev(sd[prefix + k], self.device)  # noqa: E731
        c = self.config
        self.w_patch = torch.zeros((1024, 592), dtype=F32, device=self.device)       # K = 588 zero-padded to 16 | 592
        self.w_patch[:, :588] = g("embeddings.patch_embedding.weight").reshape(1024, 588)
        self.cls, self.pos = g("embeddings.class_embedding"), g("embeddings.position_embedding.weight")
        self.pre_g, self.pre_b = g("pre_layrnorm.weight"), g("pre_layrnorm.bias")
        self.layers = []
        for i in range(c.num_hidden_layers):
            p = f"encoder.layers.{i}."
            if prefix + p + "layer_norm1.weight" not in sd:
                break
            L = {"ln1_g": g(p + "layer_norm1.weight"), "ln1_b": g(p + "layer_norm1.bias"),
                 "ln2_g": g(p + "layer_norm2.weight"), "ln2_b": g(p + "layer_norm2.bias"),
                 "w_qkv": torch.cat([g(p + f"self_attn.{n}_proj.weight") for n in "qkv"], 0).contiguous(),
                 "b_qkv": torch.cat([g(p + f"self_attn.{n}_proj.bias") for n in "qkv"], 0).contiguous(),
                 "w_o": g(p + "self_attn.out_proj.weight"), "b_o": g(p + "self_attn.out_proj.bias"),
                 "w_fc1": g(p + "mlp.fc1.weight"), "b_fc1": g(p + "mlp.fc1.bias"),
                 "w_fc2": g(p + "mlp.fc2.weight"), "b_fc2": g(p + "mlp.fc2.bias")}
            self.layers.append(L)
        if self.layers and self.layers[0]["w_fc1"].shape[0] != c.intermediate_size:
            c.intermediate_size = int(self.layers[0]["w_fc1"].shape[0])
        if len(self.layers) < c.num_hidden_layers:
            if shallow_ok:
                c.num_hidden_layers = len(self.layers)
            elif not (truncated_ok or getattr(c, "truncated_ok", False)):
                raise ValueError(f"vision tower state dict holds {len(self.layers)} encoder layers, config says {c.num_hidden_layers}")
        self.loaded = True
        return self

    def init_random(self, seed: int = 0, layers: Optional[int] = None) -> "PreciseCLIPVisionTower":
        """Random fp32 weights generated on the device (bench only: what the 1e-3 mode costs in throughput)."""
        c = self.config
        g = torch.Generator(device=self.device).manual_seed(seed)
        rn = lambda shape, std: torch.randn(shape, generator=g, device=self.device, dtype=F32) * std  # noqa: E731
        H, I = c.hidden_size, c.intermediate_size
        self.w_patch = torch.zeros((1024, 592), dtype=F32, device=self.device)
        self.w_patch[:, :588] = rn((1024, 588), 0.02)
        self.cls, self.pos = rn((H,), H ** -0.5), rn((257, H), 0.02)
        self.pre_g, self.pre_b = torch.ones(H, device=self.device), torch.zeros(H, device=self.device)
        in_std = H ** -0.5 * (2 * c.num_hidden_layers) ** -0.5
        one, zero = torch.ones(H, device=self.device), torch.zeros(H, device=self.device)
        self.layers = [dict(ln1_g=one, ln1_b=zero, ln2_g=one, ln2_b=zero, w_qkv=rn((3 * H, H), 0.02), b_qkv=rn((3 * H,), 0.02),
                            w_o=rn((H, H), H ** -0.5), b_o=rn((H,), 0.02), w_fc1=rn((I, H), (2 * H) ** -0.5), b_fc1=rn((I,), 0.02),
                            w_fc2=rn((H, I), in_std * 2), b_fc2=rn((H,), 0.02))
                       for _ in range(c.num_hidden_layers if layers is None else layers)]
        self.loaded = True
        return self

    def n_layers_for(self, select_layer: int) -> int:
        n = self.config.num_hidden_layers
        idx = select_layer if select_layer >= 0 else n + 1 + select_layer
        if not 0 <= idx <= n:
            raise IndexError(f"select_layer {select_layer} out of range for {n} layers")
        return idx

    def encode(self, frames: torch.Tensor, select_layer: int = -2, chunk: int = 128, keep_all: bool = False):
        """frames [F,3,224,224] -> fp32 [F,257,1024] = hidden_states[select_layer]."""
        if not self.loaded:
            raise RuntimeError("vision tower has no weights")
        if frames.dim() != 4 or tuple(frames.shape[1:]) != (3, 224, 224):
            raise ValueError(f"Input image size ({tuple(frames.shape)}) doesn't match model (3*224*224).")
        frames = frames.to(device=self.device, dtype=F32).contiguous()
        nl = self.n_layers_for(select_layer)
        if nl > len(self.layers):
            raise RuntimeError(f"need {nl} encoder layers, tower holds {len(self.layers)}")
        eps = self.config.layer_norm_eps
        outs, states = [], [[] for _ in range(nl + 1)]
        with runtime.stream_lock():
            for f0 in range(0, frames.shape[0], chunk):
                fr = frames[f0:f0 + chunk]
                Fn = fr.shape[0]
                patch = F.gemm(F.patchify(fr), self.w_patch)                                   # [F*256, 1024]
                h = ops.vit_embed_ln(patch, self.cls, self.pos, self.pre_g, self.pre_b, Fn, eps)   # fp32 kernel already
                if keep_all:
                    states[0].append(h.clone())
                for li, L in enumerate(self.layers[:nl]):
                    x = F.norm_for_gemm(h, L["ln1_g"], L["ln1_b"], eps, 3072)
                    att = F.vit_attention(F.gemm(x, L["w_qkv"], L["b_qkv"]), Fn)
                    F.gemm(att, L["w_o"], L["b_o"], residual=h, out=h)
                    x = F.norm_for_gemm(h, L["ln2_g"], L["ln2_b"], eps, L["w_fc1"].shape[0])
                    mid = F.gemm(x, L["w_fc1"], L["b_fc1"], epilogue=ops.EPI_QUICK_GELU)
                    F.gemm(mid, L["w_fc2"], L["b_fc2"], residual=h, out=h)
                    if keep_all:
                        states[li + 1].append(h.clone())
                outs.append(h)
        res = torch.cat(outs, 0).view(-1, 257, 1024)
        if keep_all:
            return res, [torch.cat(s, 0).view(-1, 257, 1024) for s in states]
        return res

    def __call__(self, pixel_values: torch.Tensor, output_hidden_states: bool = True):
        from types import SimpleNamespace
        last, states = self.encode(pixel_values, select_layer=len(self.layers), keep_all=True)
        return SimpleNamespace(last_hidden_state=last, hidden_states=tuple(states))

    def requires_grad_(self, flag: bool = False):
        return self

    def to(self, *a, **k):
        return self


class F32KVCache:
    """fp32 KV cache with the interface of llama.HipKVCache (legacy tuple indexing + get_seq_length)."""

    def __init__(self, layers: int, batch: int, heads: int, ctx_max: int, device):
        self.k = [torch.zeros((batch, heads, ctx_max, 128), dtype=F32, device=device) for _ in range(layers)]
        self.v = [torch.zeros((batch, heads, ctx_max, 128), dtype=F32, device=device) for _ in range(layers)]
        self.seq_len, self.ctx_max, self.batch = 0, ctx_max, batch
        self.key_valid: Optional[torch.Tensor] = None
        self.generation, self.growable, self.limit = 0, False, ctx_max

    def reserve(self, n: int) -> None:
        """Same contract as llama.HipKVCache.reserve."""
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

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.seq_len

    def __len__(self):
        return len(self.k)

    def __bool__(self):
        return self.seq_len > 0

    def __getitem__(self, layer: int):
        return (self.k[layer][:, :, :self.seq_len], self.v[layer][:, :, :self.seq_len])

    def __iter__(self):
        for i in range(len(self.k)):
            yield self[i]


class PreciseLlama:
    """hf:llama/modeling_llama.py:53-417 (RMSNorm, rotate-half RoPE, causal + padding attention, SwiGLU) in fp32."""
    precision = "fp32"

    def __init__(self, hidden: int, heads: int, intermediate: int, layers: int, vocab: int, eps: float, rope_theta: float = 10000.0,
                 max_positions: int = 2048, device="cuda:0", pack_weights=None):
        if hidden != heads * 128:
            raise ValueError("HIP Llama path requires head_dim == 128 (hidden = heads*128)")
        if hidden % 16 or intermediate % 16:
            raise ValueError("hidden and intermediate sizes must be multiples of 16")
        self.H, self.heads, self.I, self.L, self.V, self.eps = hidden, heads, intermediate, layers, vocab, eps
        self.Vpad = (vocab + 3) // 4 * 4
        self.device = torch.device(device)
        self.max_positions = max_positions
        inv = 1.0 / (rope_theta ** (torch.arange(0, 128, 2, dtype=F32) / 128))                   # hf:llama 95-124
        ang = torch.arange(max_positions, dtype=F32)[:, None] * inv[None]
        self.cos, self.sin = ang.cos().contiguous().to(self.device), ang.sin().contiguous().to(self.device)
        self.layers: List[Dict[str, torch.Tensor]] = []
        self.loaded = False

    def load_state_dict(self, sd: Dict) -> "PreciseLlama":
        d = self.device
        self.embed = _dev(sd["model.embed_tokens.weight"], d)
        self.layers = []
        for i in range(self.L):
            p = f"model.layers.{i}."
            gate, up = _dev(sd[p + "mlp.gate_proj.weight"], d), _dev(sd[p + "mlp.up_proj.weight"], d)
            self.layers.append({
                "ln1": _dev(sd[p + "input_layernorm.weight"], d), "ln2": _dev(sd[p + "post_attention_layernorm.weight"], d),
                "w_qkv": torch.cat([_dev(sd[p + f"self_attn.{n}_proj.weight"], d) for n in "qkv"], 0).contiguous(),
                "w_o": _dev(sd[p + "self_attn.o_proj.weight"], d),
                "w_gu": torch.stack([gate, up], dim=1).reshape(2 * self.I, self.H).contiguous(),   # (gate_j, up_j) interleaved
                "w_down": _dev(sd[p + "mlp.down_proj.weight"], d)})
        self.norm = _dev(sd["model.norm.weight"], d)
        self.lm_head = torch.zeros((self.Vpad, self.H), dtype=F32, device=d)
        self.lm_head[:self.V] = _dev(sd["lm_head.weight"], d)
        self.loaded = True
        return self

    def init_random(self, seed: int = 0, std: float = 0.02) -> "PreciseLlama":
        """Random fp32 weights generated on the device (bench only)."""
        d = self.device
        g = torch.Generator(device=d).manual_seed(seed)
        rn = lambda shape: torch.randn(shape, generator=g, device=d, dtype=F32) * std  # noqa: E731
        self.embed = rn((self.V, self.H))
        self.layers = [dict(ln1=torch.ones(self.H, device=d), ln2=torch.ones(self.H, device=d), w_qkv=rn((3 * self.H, self.H)),
                            w_o=rn((self.H, self.H)), w_gu=rn((2 * self.I, self.H)), w_down=rn((self.H, self.I))) for _ in range(self.L)]
        self.norm = torch.ones(self.H, device=d)
        self.lm_head = torch.zeros((self.Vpad, self.H), dtype=F32, device=d)
        self.lm_head[:self.V] = rn((self.V, self.H))
        self.loaded = True
        return self

    def new_cache(self, batch: int, ctx_max: Optional[int] = None) -> F32KVCache:
        return F32KVCache(self.L, batch, self.heads, ctx_max or self.max_positions, self.device)

    def forward(self, h: torch.Tensor, B: int, S: int, cache: F32KVCache, n_layers: Optional[int] = None,
                collect: Optional[list] = None, attn: Optional[list] = None) -> torch.Tensor:
        """h fp32 [B*S, H] (modified in place) -> final-norm hidden fp32 [B*S, H]; appends S positions to ``cache``;
        ``collect`` as in HipLlama.forward (HF's output_hidden_states)."""
        if not self.loaded:
            raise RuntimeError("Llama engine has no weights")
        if not isinstance(cache, F32KVCache):
            raise TypeError("the fp32 engine needs the F32KVCache it created")
        past = cache.seq_len
        cache.reserve(past + S)
        if cache.batch != B:
            raise ValueError("cache batch mismatch")
        with runtime.stream_lock():
            nl = self.L if n_layers is None else n_layers
            if collect is not None:
                collect.append(h.clone())
            for li in range(nl):
                L = self.layers[li]
                qkv = F.gemm(F.norm_for_gemm(h, L["ln1"], None, self.eps, 3 * self.H), L["w_qkv"])
                F.rope_kv(qkv, cache.k[li], cache.v[li], self.cos, self.sin, B, S, self.heads, past)
                att = F.llama_attention(qkv, cache.k[li], cache.v[li], cache.key_valid, B, S, self.heads, past)
                if attn is not None:                        # output_attentions (as HipLlama.forward): rotated q, rotated K cache
                    from . import ops as _ops
                    attn.append(_ops.attention_probs(qkv, cache.k[li], cache.key_valid, B, S, self.heads, past))
                F.gemm(att, L["w_o"], residual=h, out=h)
                mid = F.gemm(F.norm_for_gemm(h, L["ln2"], None, self.eps, 2 * self.I), L["w_gu"], epilogue=ops.EPI_SWIGLU)
                F.gemm(mid, L["w_down"], residual=h, out=h)
                if collect is not None and li + 1 < nl:
                    collect.append(h.clone())
            cache.seq_len = past + S
            out = F.norm(h, self.norm, None, self.eps)
            if collect is not None:
                collect.append(out.clone())
            return out

    def logits(self, x: torch.Tensor) -> torch.Tensor:
        return F.gemm(x, self.lm_head)[:, :self.V]
