"""CPU oracle for Valley's visual-token hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 PyTorch ops on the CPU, the arithmetic the reference executes on
its hot path, so that the HIP path can be checked against it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; nothing under
``valley_amd/`` does, and the product path fails loudly when the HIP library is missing.

What it restates (citations are into /root/reference unless they start with ``hf:``, which means
the installed third-party ``transformers`` 5.15.0 under
/usr/local/lib/python3.10/dist-packages/transformers/models/ — the reference pins
``transformers @ git+...@cae78c46`` in pyproject.toml:19 and does not vendor it):

* CLIP ViT-L/14 tower to ``hidden_states[select_layer]``
  - embeddings: hf:clip/modeling_clip.py:203-218 (patch conv, CLS concat, position add)
  - pre_layrnorm + encoder layers: hf:clip/modeling_clip.py:641-647, 353-383
  - attention (non-causal, fp32 softmax, scale hd^-0.5): hf:clip/modeling_clip.py:258-277, 302-334
  - MLP with quick_gelu x*sigmoid(1.702x): hf:clip/modeling_clip.py:338-350
  - called from valley/model/valley_model.py:172-183 (per clip, hidden_states[-2], CLS kept)
* mm_projector + temporal pooling + CLS pick: valley/model/valley_model.py:187-193, 206-215,
  v2 importance pooling :113-121, v3 transformer-delta :123-133
* visual-token splice with the reference's error behaviour: valley/model/valley_model.py:195-247
* Llama decoder (RMSNorm / RoPE rotate-half / causal+padding softmax attention / SwiGLU):
  hf:llama/modeling_llama.py:51-67 (RMSNorm), 73-124 + 127-157 (RoPE), 160-173 (MLP),
  191-213 (eager attention), 217-289 (attention block), 292-332 (decoder layer)
  called from valley/model/valley_model.py:249-254; lm_head :304-305
* manual prefill + KV decode loop and sampling rule: valley/serve/model_worker.py:371-394

Pinning: tests/test_oracle_golden.py checks every function here against fixtures under
tests/golden/ that tools/gen_goldens.py captured in the authoring container by importing the
reference itself (valley.model.valley_model, with stub modules for its absent I/O dependencies)
on top of the installed transformers, fed with valley_amd.weights' deterministic tensors.  The
reference has no tests or golden vectors of its own, so those captured outputs are
the anchor; tolerance fp32 vs fp32 is 2e-5 max-abs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# configs
# --------------------------------------------------------------------------------------------
@dataclass
class VisionCfg:
    hidden: int = 1024
    heads: int = 16
    intermediate: int = 4096
    layers: int = 24
    image_size: int = 224
    patch: int = 14
    eps: float = 1e-5


@dataclass
class LlamaCfg:
    hidden: int = 4096
    heads: int = 32
    intermediate: int = 11008
    layers: int = 32
    vocab: int = 32006
    eps: float = 1e-5           # 1e-5 Llama-2, 1e-6 LLaMA-1/Vicuna
    rope_theta: float = 10000.0


@dataclass
class TokenIds:
    """The six ids the entry points bind onto vision_tower.config
    (valley/inference/run_valley.py:13-18)."""
    im_patch_token: int
    vi_frame_token: int
    im_start_token: int
    im_end_token: int
    vi_start_token: int
    vi_end_token: int


def _t(w: Dict, k: str) -> torch.Tensor:
    v = w[k]
    if not isinstance(v, torch.Tensor):
        v = torch.from_numpy(v)
    return v.float()


# --------------------------------------------------------------------------------------------
# optional "same-dtype" mode (the bf16-rounded-at-the-same-points oracle)
# --------------------------------------------------------------------------------------------
# The reference arithmetic above is fp32.  The HIP production path stores GEMM operands and most activations in bf16
# (fp32 accumulation, fp32 softmax / norm statistics, fp32 residual stream — DESIGN.md §3).  Inside ``with rounding():``
# every function of this file rounds to bf16 exactly where that pipeline stores bf16: GEMM weights (``_tw``) and the
# tensors marked ``_q(...)`` below.  What remains between the HIP path and this mode is summation order (and the
# flash-style online softmax), so the HIP path can be held to ~1e-3-level agreement with it, while its distance to the
# fp32 mode measures the cost of bf16 storage itself.  Norm parameters, biases, position / class embeddings and RoPE
# tables stay fp32 in both.
_ROUND = False
_ROUND_DTYPE = torch.bfloat16
_ROUND_KEEP = frozenset()         # storage points exempt from the rounding (tools/logit_precision_study.py switches them off one by one)
# every storage point of the pipeline, by name (the ``site`` of each ``_q``): which tensor the HIP path holds in 16 bits
ROUND_SITES = ("weights", "vit.pixels", "vit.ln", "vit.qkv", "vit.probs", "vit.attn_out", "vit.o_out", "vit.gelu", "vit.fc2_out",
               "proj.in", "proj.rows", "embed", "ll.ln", "ll.qkv", "ll.rope", "ll.probs", "ll.attn_out", "ll.o_out", "ll.swiglu",
               "ll.down_out", "ll.final_norm")


class rounding:
    """Context manager: ``with oracle.rounding(): ...`` evaluates the path with 16-bit storage rounding — bf16 by default,
    ``rounding(torch.float16)`` for the fp16 library (VALLEY_PRECISION=fp16, the reference's own inference dtype).
    ``keep``: names from ROUND_SITES that stay fp32 (the ablation of tools/logit_precision_study.py)."""

    def __init__(self, dtype: torch.dtype = torch.bfloat16, keep=()):
        self._dtype = dtype
        self._keep = frozenset(keep)
        assert self._keep <= set(ROUND_SITES), self._keep - set(ROUND_SITES)

    def __enter__(self):
        global _ROUND, _ROUND_DTYPE, _ROUND_KEEP
        self._old, _ROUND = (_ROUND, _ROUND_DTYPE, _ROUND_KEEP), True
        _ROUND_DTYPE, _ROUND_KEEP = self._dtype, self._keep
        return self

    def __exit__(self, *exc):
        global _ROUND, _ROUND_DTYPE, _ROUND_KEEP
        _ROUND, _ROUND_DTYPE, _ROUND_KEEP = self._old
        return False


def _q(x: torch.Tensor, site: str = "") -> torch.Tensor:
    """Round to the storage type (nearest-even) and back when the same-dtype mode is on; identity otherwise."""
    return x.to(_ROUND_DTYPE).float() if _ROUND and site not in _ROUND_KEEP else x


def _tw(w: Dict, k: str) -> torch.Tensor:
    """A GEMM weight (kept in bf16 by the HIP path)."""
    return _q(_t(w, k), "weights")


# --------------------------------------------------------------------------------------------
# CLIP vision tower
# --------------------------------------------------------------------------------------------
def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """hf:activations.py QuickGELUActivation: x * sigmoid(1.702 x)."""
    return x * torch.sigmoid(1.702 * x)


def clip_embeddings(pixels: torch.Tensor, w: Dict, cfg: VisionCfg, prefix: str = "") -> torch.Tensor:
    """hf:clip/modeling_clip.py:203-218.  pixels [F,3,H,W] -> [F, 1+P, D]."""
    pw = _tw(w, prefix + "embeddings.patch_embedding.weight")
    x = F.conv2d(_q(pixels.float(), "vit.pixels"), pw, bias=None, stride=cfg.patch)      # [F, D, g, g]
    x = x.flatten(2).transpose(1, 2)                                        # [F, P, D]
    cls = _t(w, prefix + "embeddings.class_embedding").expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1)
    return x + _t(w, prefix + "embeddings.position_embedding.weight")[None]


def clip_attention(x: torch.Tensor, w: Dict, p: str, heads: int) -> torch.Tensor:
    """hf:clip/modeling_clip.py:302-334 with eager_attention_forward :258-277 (no mask)."""
    Fn, N, D = x.shape
    hd = D // heads
    q = _q(F.linear(x, _tw(w, p + "q_proj.weight"), _t(w, p + "q_proj.bias")), "vit.qkv").view(Fn, N, heads, hd).transpose(1, 2)
    k = _q(F.linear(x, _tw(w, p + "k_proj.weight"), _t(w, p + "k_proj.bias")), "vit.qkv").view(Fn, N, heads, hd).transpose(1, 2)
    v = _q(F.linear(x, _tw(w, p + "v_proj.weight"), _t(w, p + "v_proj.bias")), "vit.qkv").view(Fn, N, heads, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
    a = _q(torch.softmax(s, dim=-1, dtype=torch.float32), "vit.probs")
    o = _q(torch.matmul(a, v).transpose(1, 2).reshape(Fn, N, D), "vit.attn_out")
    return _q(F.linear(o, _tw(w, p + "out_proj.weight"), _t(w, p + "out_proj.bias")), "vit.o_out")


def clip_mlp(h: torch.Tensor, w: Dict, p: str) -> torch.Tensor:
    """hf:clip/modeling_clip.py:338-350: fc2(quick_gelu(fc1(x))); p ends in 'mlp.'."""
    h = F.linear(h, _tw(w, p + "fc1.weight"), _t(w, p + "fc1.bias"))
    return _q(F.linear(_q(quick_gelu(h), "vit.gelu"), _tw(w, p + "fc2.weight"), _t(w, p + "fc2.bias")), "vit.fc2_out")


def clip_layer(x: torch.Tensor, w: Dict, p: str, cfg: VisionCfg) -> torch.Tensor:
    """hf:clip/modeling_clip.py:353-383 (pre-LN encoder layer)."""
    h = _q(F.layer_norm(x, (cfg.hidden,), _t(w, p + "layer_norm1.weight"), _t(w, p + "layer_norm1.bias"), cfg.eps), "vit.ln")
    x = x + clip_attention(h, w, p + "self_attn.", cfg.heads)
    h = _q(F.layer_norm(x, (cfg.hidden,), _t(w, p + "layer_norm2.weight"), _t(w, p + "layer_norm2.bias"), cfg.eps), "vit.ln")
    return x + clip_mlp(h, w, p + "mlp.")


def clip_hidden_states(pixels: torch.Tensor, w: Dict, cfg: VisionCfg, n_layers: Optional[int] = None,
                       prefix: str = "") -> List[torch.Tensor]:
    """All encoder hidden states, HF convention: [0] = pre_layrnorm(embeddings),
    [i] = output of layer i (hf:clip/modeling_clip.py:641-647)."""
    x = clip_embeddings(pixels, w, cfg, prefix)
    x = F.layer_norm(x, (cfg.hidden,), _t(w, prefix + "pre_layrnorm.weight"), _t(w, prefix + "pre_layrnorm.bias"), cfg.eps)
    hs = [x]
    L = cfg.layers if n_layers is None else n_layers
    for i in range(L):
        x = clip_layer(x, w, prefix + f"encoder.layers.{i}.", cfg)
        hs.append(x)
    return hs


def vit_select(pixels: torch.Tensor, w: Dict, cfg: VisionCfg, select_layer: int = -2, prefix: str = "") -> torch.Tensor:
    """valley_model.py:180-183: hidden_states[select_layer][:, :] (CLS kept).  Only the layers that
    contribute are evaluated: hidden_states has cfg.layers+1 entries, so index -2 is the output of
    layer cfg.layers-1 and the last layer / post_layernorm never run."""
    idx = select_layer if select_layer >= 0 else cfg.layers + 1 + select_layer
    return clip_hidden_states(pixels, w, cfg, n_layers=idx, prefix=prefix)[idx]


# --------------------------------------------------------------------------------------------
# projector + temporal pooling
# --------------------------------------------------------------------------------------------
def mm_project(feats: torch.Tensor, w: Dict) -> torch.Tensor:
    """valley_model.py:54-55,190: Linear(mm_hidden -> H) + bias on every token."""
    return F.linear(_q(feats, "proj.in"), _tw(w, "model.mm_projector.weight"), _t(w, "model.mm_projector.bias"))


def sinusoid_position_matrix(seq_len: int, d: int, n: float = 10000.0) -> torch.Tensor:
    """valley_model.py:104-111 (vectorised; same values)."""
    k = torch.arange(seq_len, dtype=torch.float32)[:, None]
    i = torch.arange(d // 2, dtype=torch.float32)[None, :]
    den = torch.pow(torch.tensor(n), 2 * i / d)
    P = torch.zeros(seq_len, d)
    P[:, 0::2] = torch.sin(k / den)
    P[:, 1::2] = torch.cos(k / den)
    return P


def pool_clip(proj: torch.Tensor, method: str, w: Optional[Dict] = None, nhead: int = 8) -> Tuple[torch.Tensor, torch.Tensor]:
    """valley_model.py:206-215 on one clip.  proj [T, 1+P, H] (already projected).
    Returns (pooled patch tokens [P, H], per-frame CLS tokens [T, H])."""
    patches = proj[:, 1:, :]
    if method == "mean":
        pooled = patches.mean(dim=0)
    elif method == "max":
        pooled = patches.max(dim=0)[0]
    elif method == "temporal_importance":
        # valley_model.py:113-121
        flat = torch.flatten(patches, start_dim=1)                          # [T, P*H]
        score = torch.softmax(F.linear(flat, _t(w, "model.pooling_layer.weight"), _t(w, "model.pooling_layer.bias")), dim=0)
        pooled = (score.unsqueeze(2) * patches).sum(dim=0)
    elif method == "temporal_transformer":
        # valley_model.py:123-133 : nn.TransformerEncoderLayer(d_model=H, nhead=8, batch_first=True),
        # torch defaults: dim_feedforward 2048, relu, post-LN (norm_first False), eps 1e-5.
        x = patches.permute(1, 0, 2)                                        # [P, T, H]
        T = x.shape[1]
        pos = _t(w, "model.position_matrix")[:T][None]
        delta = torch_encoder_layer(x + pos, w, "model.transformer_delta_encoder.layers.0.", nhead)[:, -1, :]
        pooled = delta + x.mean(dim=1)
    else:
        raise ValueError(method)
    return pooled, proj[:, 0, :]


def torch_encoder_layer(x: torch.Tensor, w: Dict, p: str, nhead: int) -> torch.Tensor:
    """torch.nn.TransformerEncoderLayer forward (post-LN, ReLU FFN), eval mode."""
    B, T, H = x.shape
    hd = H // nhead
    qkv = F.linear(x, _t(w, p + "self_attn.in_proj_weight"), _t(w, p + "self_attn.in_proj_bias"))
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.view(B, T, nhead, hd).transpose(1, 2)
    k = k.view(B, T, nhead, hd).transpose(1, 2)
    v = v.view(B, T, nhead, hd).transpose(1, 2)
    a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, H)
    o = F.linear(o, _t(w, p + "self_attn.out_proj.weight"), _t(w, p + "self_attn.out_proj.bias"))
    x = F.layer_norm(x + o, (H,), _t(w, p + "norm1.weight"), _t(w, p + "norm1.bias"), 1e-5)
    f = F.linear(F.relu(F.linear(x, _t(w, p + "linear1.weight"), _t(w, p + "linear1.bias"))),
                 _t(w, p + "linear2.weight"), _t(w, p + "linear2.bias"))
    return F.layer_norm(x + f, (H,), _t(w, p + "norm2.weight"), _t(w, p + "norm2.bias"), 1e-5)


# --------------------------------------------------------------------------------------------
# visual-token splice (integer index logic + row copies)
# --------------------------------------------------------------------------------------------
def splice_visual_tokens(input_ids: torch.Tensor, inputs_embeds: torch.Tensor,
                         image_features: Sequence[torch.Tensor], tok: TokenIds, method: str = "mean",
                         w: Optional[Dict] = None, pooled: Optional[Sequence[Tuple[torch.Tensor, torch.Tensor]]] = None) -> torch.Tensor:
    """valley_model.py:195-247, statement for statement.

    image_features: per multimodal sample, projected features [T_i, 1+P, H].  ``cur_image_idx``
    advances only on samples that contain <im_patch> (:198-202,246).  Raises ValueError exactly where
    the reference does (:219-220, :226-227); any failure in the <vi_start> block is swallowed and
    the image-only splice kept (:231-244).
    """
    new_embeds = []
    cur_image_idx = 0
    for ids, emb in zip(input_ids, inputs_embeds):
        if (ids == tok.im_patch_token).sum() == 0:
            new_embeds.append(emb)                       # + 0*dummy.sum() is a numeric no-op (:200)
            continue
        feats = image_features[cur_image_idx]
        pooled_c, cls = pool_clip(feats, method, w) if pooled is None else pooled[cur_image_idx]   # (same-dtype mode: pre-pooled rows)
        P = pooled_c.shape[0]
        if (ids == tok.im_start_token).sum() != (ids == tok.im_end_token).sum():
            raise ValueError("The number of im_start_token and im_end_token should be the same")
        cur = emb.clone()
        for pos in torch.where(ids == tok.im_start_token)[0]:
            pos = int(pos)
            # reference indexes ids[pos+P+1] unguarded: IndexError if the prompt is too short
            if ids[pos + P + 1] != tok.im_end_token:
                raise ValueError("Seems that the image is cut.")
            cur = torch.cat((cur[:pos + 1], pooled_c, cur[pos + P + 1:]), dim=0)
        try:
            if (ids == tok.vi_start_token).sum() != (ids == tok.vi_end_token).sum():
                raise ValueError("The number of vi_start_token and vi_end_token should be the same")
            T = cls.shape[0]
            assert (ids == tok.vi_frame_token).sum() == T
            vid = cur.clone()
            for pos in torch.where(ids == tok.vi_start_token)[0]:
                pos = int(pos)
                if ids[pos + T + 1] != tok.vi_end_token:
                    raise ValueError("Seems that the image is cut.")
                vid = torch.cat((vid[:pos + 1], cls, vid[pos + T + 1:]), dim=0)
        except Exception:                                # bare except in the reference (:243)
            vid = cur.clone()
        new_embeds.append(vid)
        cur_image_idx += 1
    return torch.stack(new_embeds, dim=0)


# --------------------------------------------------------------------------------------------
# Llama decoder
# --------------------------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """hf:llama/modeling_llama.py:61-66."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return weight * (x.float() * torch.rsqrt(var + eps))


def rope_cos_sin(positions: torch.Tensor, head_dim: int, theta: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """hf:llama/modeling_llama.py:95-124: inv_freq = theta^(-2i/d); emb = cat(freqs, freqs)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = positions.float()[..., None] * inv                               # [..., d/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """hf:llama/modeling_llama.py:127-131."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """hf:llama/modeling_llama.py:134-157, x [B,h,S,d], cos/sin [B,S,d]."""
    return x * cos[:, None] + rotate_half(x) * sin[:, None]


def build_additive_mask(attention_mask: Optional[torch.Tensor], B: int, q_len: int, kv_len: int) -> torch.Tensor:
    """causal AND padding as an additive fp32 mask [B,1,q,kv] (hf masking_utils semantics:
    key j visible to query i iff j <= i + past and attention_mask[b, j] == 1)."""
    past = kv_len - q_len
    i = torch.arange(q_len)[:, None] + past
    j = torch.arange(kv_len)[None, :]
    allowed = (j <= i)[None, None].expand(B, 1, q_len, kv_len)
    if attention_mask is not None:
        allowed = allowed & attention_mask[:, None, None, :kv_len].bool()
    return torch.where(allowed, 0.0, torch.finfo(torch.float32).min)


def llama_attention_block(h: torch.Tensor, w: Dict, p: str, cfg: LlamaCfg, cos, sin, mask,
                          past: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, attn_out: Optional[list] = None):
    """hf:llama/modeling_llama.py:217-289 (LlamaAttention.forward, eager): q/k/v projections, RoPE, KV concat,
    softmax(QK^T * hd^-0.5 + mask) V in fp32, o_proj.  h [B,S,H] is the NORMED hidden state; p ends in 'self_attn.'."""
    B, S, H = h.shape
    hd = H // cfg.heads
    q = _q(F.linear(h, _tw(w, p + "q_proj.weight")), "ll.qkv").view(B, S, cfg.heads, hd).transpose(1, 2)
    k = _q(F.linear(h, _tw(w, p + "k_proj.weight")), "ll.qkv").view(B, S, cfg.heads, hd).transpose(1, 2)
    v = _q(F.linear(h, _tw(w, p + "v_proj.weight")), "ll.qkv").view(B, S, cfg.heads, hd).transpose(1, 2)
    q = _q(apply_rope(q, cos, sin), "ll.rope")
    k = _q(apply_rope(k, cos, sin), "ll.rope")
    if past is not None:
        k = torch.cat([past[0], k], dim=2)
        v = torch.cat([past[1], v], dim=2)
    s = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5) + mask
    a = _q(torch.softmax(s, dim=-1, dtype=torch.float32), "ll.probs")
    if attn_out is not None:                                 # HF's ``output_attentions``: the probabilities [B, heads, S, kv_len]
        attn_out.append(a)
    o = _q(torch.matmul(a, v).transpose(1, 2).reshape(B, S, H), "ll.attn_out")
    out = F.linear(o, _tw(w, p + "o_proj.weight"))
    # prefill hands the projection to the residual add as a bf16 tensor; the <= 8-row decode step adds in the GEMV epilogue
    return (_q(out, "ll.o_out") if B * S > 8 else out), (k, v)


def llama_mlp(h: torch.Tensor, w: Dict, p: str) -> torch.Tensor:
    """hf:llama/modeling_llama.py:160-173: down(silu(gate(x)) * up(x)); p ends in 'mlp.'."""
    g = F.linear(h, _tw(w, p + "gate_proj.weight"))
    u = F.linear(h, _tw(w, p + "up_proj.weight"))
    out = F.linear(_q(F.silu(g) * u, "ll.swiglu"), _tw(w, p + "down_proj.weight"))
    return _q(out, "ll.down_out") if h.shape[0] * h.shape[1] > 8 else out


def llama_layer(x: torch.Tensor, w: Dict, p: str, cfg: LlamaCfg, cos, sin, mask,
                past: Optional[Tuple[torch.Tensor, torch.Tensor]], attn_out: Optional[list] = None):
    """hf:llama/modeling_llama.py:292-332 (pre-norm decoder layer)."""
    h = _q(rms_norm(x, _t(w, p + "input_layernorm.weight"), cfg.eps), "ll.ln")
    o, kv = llama_attention_block(h, w, p + "self_attn.", cfg, cos, sin, mask, past, attn_out)
    x = x + o
    h = _q(rms_norm(x, _t(w, p + "post_attention_layernorm.weight"), cfg.eps), "ll.ln")
    return x + llama_mlp(h, w, p + "mlp."), kv


def llama_forward(inputs_embeds: torch.Tensor, w: Dict, cfg: LlamaCfg,
                  attention_mask: Optional[torch.Tensor] = None,
                  past: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None,
                  n_layers: Optional[int] = None, attn_out: Optional[list] = None):
    """hf:llama/modeling_llama.py:372-417: positions = arange(S) + past_len (independent of the
    padding mask), final RMSNorm.  Returns (hidden [B,S,H], new past).  ``attn_out`` (a list) receives every layer's
    attention probabilities (HF's ``output_attentions``, valley_model.py:281,324-330)."""
    B, S, H = inputs_embeds.shape
    past_len = 0 if past is None else past[0][0].shape[2]
    pos = (torch.arange(S) + past_len)[None].expand(B, S)
    cos, sin = rope_cos_sin(pos, H // cfg.heads, cfg.rope_theta)
    mask = build_additive_mask(attention_mask, B, S, past_len + S)
    x = inputs_embeds.float()
    new_past = []
    L = cfg.layers if n_layers is None else n_layers
    for i in range(L):
        x, kv = llama_layer(x, w, f"model.layers.{i}.", cfg, cos, sin, mask, None if past is None else past[i], attn_out)
        new_past.append(kv)
    return _q(rms_norm(x, _t(w, "model.norm.weight"), cfg.eps), "ll.final_norm"), new_past


# --------------------------------------------------------------------------------------------
# whole path
# --------------------------------------------------------------------------------------------
def valley_forward(input_ids: torch.Tensor, images, w: Dict, vw: Dict, lcfg: LlamaCfg, vcfg: VisionCfg,
                   tok: TokenIds, attention_mask: Optional[torch.Tensor] = None, past=None,
                   select_layer: int = -2, method: str = "mean", vprefix: str = "", attn_out: Optional[list] = None):
    """ValleyLlamaForCausalLM.forward (valley_model.py:272-330) without the loss.
    images: [B,T,3,H,W] tensor or list of [T_i,3,H,W] (valley_model.py:168-184)."""
    emb = F.embedding(input_ids, _tw(w, "model.embed_tokens.weight"))
    if images is not None and input_ids.shape[1] != 1:
        raw = [vit_select(clip, vw, vcfg, select_layer, vprefix) for clip in images]
        if _ROUND and method in ("mean", "max"):
            # the HIP path's storage points around the projector: mean pools the fp32 tower output first and projects
            # 256 + T bf16 rows; max projects every (bf16) token into fp32 and pools there; visual tokens are bf16
            pooled = []
            for f in raw:
                if method == "mean":
                    rows = _q(torch.cat([f[:, 1:].mean(0), f[:, 0]], 0), "proj.in")
                    rows = _q(F.linear(rows, _tw(w, "model.mm_projector.weight"), _t(w, "model.mm_projector.bias")), "proj.rows")
                else:
                    pr = mm_project(f, w)
                    rows = _q(torch.cat([pr[:, 1:].max(0)[0], pr[:, 0]], 0), "proj.rows")
                pooled.append((rows[:f.shape[1] - 1], rows[f.shape[1] - 1:]))
            emb = splice_visual_tokens(input_ids, emb, [mm_project(f, w) for f in raw], tok, method, w, pooled=pooled)
        else:
            emb = splice_visual_tokens(input_ids, emb, [mm_project(f, w) for f in raw], tok, method, w)
    hidden, new_past = llama_forward(emb, w, lcfg, attention_mask, past, attn_out=attn_out)
    logits = F.linear(hidden, _tw(w, "lm_head.weight"))
    return logits, new_past, emb


def greedy_decode(input_ids: torch.Tensor, images, w, vw, lcfg, vcfg, tok, steps: int, **kw):
    """valley/serve/model_worker.py:371-394 with temperature < 1e-4 (argmax):
    prefill once, then feed one token at a time with the growing KV cache and an all-ones mask."""
    logits, past, _ = valley_forward(input_ids, images, w, vw, lcfg, vcfg, tok, **kw)
    out_tokens, out_logits = [], []
    for _ in range(steps):
        last = logits[:, -1, :]
        out_logits.append(last)
        token = last.argmax(dim=-1)
        out_tokens.append(token)
        logits, past, _ = valley_forward(token[:, None], None, w, vw, lcfg, vcfg, tok, past=past)
    return torch.stack(out_tokens, 1), torch.stack(out_logits, 1)


# --------------------------------------------------------------------------------------------
# the worker's streaming loop
# --------------------------------------------------------------------------------------------
# valley/util/config.py:5-13
IMAGE_PATCH, IM_START, IM_END = "<im_patch>", "<im_start>", "<im_end>"
VIDEO, VI_FRAME, VI_START, VI_END = "<video>", "<vi_frame>", "<vi_start>", "<vi_end>"


def generate_video_stream(params: dict, tokenizer, video: Optional[torch.Tensor], w, vw, lcfg, vcfg, tok: TokenIds,
                          mm_use_im_start_end: bool = False, stream_interval: int = 2, context_len: int = 2048,
                          sampler=None, trace: Optional[list] = None):
    """valley/serve/model_worker.py:321-426, statement for statement, on the oracle forward.  ``video``: what the
    worker's load_video returns, [3,T,224,224]; ``sampler(probs) -> int`` stands for ``torch.multinomial(probs, 1)``
    (:393-394) so that a test can seed it.  Yields the ``json\0`` chunks."""
    import json
    prompt = params["prompt"]
    ori_prompt = prompt
    images = None
    if video is not None:
        assert 1 == prompt.count(VIDEO), "Number of video does not match number of <video> tokens in prompt"
        frames = video.permute(1, 0, 2, 3)                                   # :337
        replace_token = IMAGE_PATCH * 256                                     # :338
        if mm_use_im_start_end:                                               # :339-340
            replace_token = IM_START + replace_token + IM_END + VI_START + VI_FRAME * frames.shape[0] + VI_END
        prompt = prompt.replace(VIDEO, replace_token)
        images = frames.unsqueeze(0)
    temperature = float(params.get("temperature", 1.0))
    max_new_tokens = min(int(params.get("max_new_tokens", 256)), 1024)
    stop_str = params.get("stop", None)
    stop_idx = None
    if stop_str is not None:
        stop_idx = tokenizer(stop_str).input_ids
        stop_idx = stop_idx[0] if len(stop_idx) == 1 else None
    input_ids = tokenizer(prompt).input_ids
    pred_ids = []
    max_src_len = context_len - max_new_tokens - 8
    input_ids = input_ids[-max_src_len:]
    past = None
    for i in range(max_new_tokens):
        if i == 0:
            logits, past, _ = valley_forward(torch.as_tensor([input_ids]), images, w, vw, lcfg, vcfg, tok)
        else:
            mask = torch.ones(1, past[0][0].shape[-2] + 1, dtype=torch.long)
            logits, past, _ = valley_forward(torch.as_tensor([[token]]), None, w, vw, lcfg, vcfg, tok, attention_mask=mask, past=past)
        last = logits[0][-1]
        if trace is not None:
            trace.append(last.clone())
        if temperature < 1e-4:
            token = int(torch.argmax(last))
        else:
            probs = torch.softmax(last / temperature, dim=-1)
            token = int(torch.multinomial(probs, num_samples=1)) if sampler is None else int(sampler(probs))
        pred_ids.append(token)
        if stop_idx is not None and token == stop_idx:
            stopped = True
        elif token == tokenizer.eos_token_id:
            stopped = True
        else:
            stopped = False
        if i % stream_interval == 0 or i == max_new_tokens - 1 or stopped:
            cur_out = tokenizer.decode(pred_ids, skip_special_tokens=True)
            pos = cur_out.rfind(stop_str)                                     # the reference passes stop_str unguarded (:408)
            if pos != -1:
                cur_out = cur_out[:pos]
                stopped = True
            yield json.dumps({"text": ori_prompt + cur_out, "error_code": 0}).encode() + b"\0"
        if stopped:
            break
