"""Drop-in for ``valley/model/valley_model.py`` (reference lines cited per symbol) on MI355X.

Same names, argument meaning and error behaviour as the reference classes; the arithmetic runs in
libvalley_hip.so through ``valley_amd.ops``.  Differences that are deliberate and documented in
DESIGN.md:

* all frames of all clips are encoded in one batch instead of the per-clip loop (:179-184);
* mean pooling happens BEFORE the projector (mean is linear: mean_T(Wx+b) = W mean_T(x) + b), which
  cuts the projector GEMM from B*T*257 to B*(256+T) rows; ``max`` pooling keeps the reference
  order (project all tokens, then max) because max does not commute;
* the dummy ``zeros(256,1024)`` projection (:192-193) and the ``0 * dummy.sum()`` add (:200) are
  numeric no-ops and are not executed;
* logits are returned in fp32 (the reference returns the model dtype).
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import ops, ops_f32, runtime
from .llama import HipKVCache, HipLlama
from .precise import F32KVCache, PreciseCLIPVisionTower, PreciseLlama
from .splice import build_row_map
from .vision_tower import HipCLIPVisionTower, VisionConfig

try:                                                     # plumbing only: config + output containers
    from transformers import LlamaConfig
    from transformers.modeling_outputs import BaseModelOutputWithPast, CausalLMOutputWithPast
except Exception:                                        # pragma: no cover - transformers is in the image
    LlamaConfig = None

    class CausalLMOutputWithPast(SimpleNamespace):
        pass

    class BaseModelOutputWithPast(SimpleNamespace):
        pass

# valley/util/config.py:1-13
IGNORE_INDEX = -100
DEFAULT_PAD_TOKEN = "[PAD]"
DEFAULT_EOS_TOKEN = "</s>"
DEFAULT_BOS_TOKEN = "</s>"
DEFAULT_UNK_TOKEN = "<unk>"
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
DEFAULT_VIDEO_TOKEN = "<video>"
DEFAULT_VIDEO_FRAME_TOKEN = "<vi_frame>"
DEFAULT_VI_START_TOKEN = "<vi_start>"
DEFAULT_VI_END_TOKEN = "<vi_end>"

if LlamaConfig is not None:
    class ValleyConfig(LlamaConfig):                     # valley_model.py:18-19
        model_type = "valley"
else:                                                    # pragma: no cover
    class ValleyConfig(SimpleNamespace):
        model_type = "valley"


def default_precision() -> str:
    """"bf16" / "fp16" (production engines: 16-bit MFMA operands in the library's storage type — bf16, or IEEE fp16 like the
    reference's own ``model.to(torch.float16)`` — fp32 accumulation and residual stream) or "fp32" (validation: every tensor
    fp32, exact f32 MFMA — valley_amd/precise.py), from VALLEY_PRECISION (read once, valley_amd/runtime.py: the 16-bit
    storage type is a property of the loaded library, not of a model object)."""
    return runtime.PRECISION


def resolve_precision(config) -> str:
    """The precision a model object is built in: ``config.valley_precision`` ("bf16" | "fp16" | "fp32") when the config names
    one, else the process default.  A 16-bit name is a REQUEST for that storage type (runtime.request_half): it selects the
    matching library while the choice is open and raises when the process is already bound to the other one — a config that
    says fp16 never runs in bf16 silently."""
    p = getattr(config, "valley_precision", None)
    if p is None:
        return default_precision()
    p = str(p).lower()
    if p not in ("bf16", "fp16", "fp32"):
        raise ValueError(f"valley_precision must be bf16, fp16 or fp32, got {p!r}")
    if p != "fp32":
        runtime.request_half(p, f"ValleyConfig.valley_precision={p!r}")
    return p


def apply_torch_dtype(config, torch_dtype, who: str) -> None:
    """``from_pretrained(..., torch_dtype=torch.float16)`` of the reference (run_valley.py:39, serve/model_worker.py:61,79):
    the caller's dtype decides the precision the model is built in — fp16 / bf16 select the library with that storage type
    (or raise when the process already holds the other one), fp32 the validation engines."""
    if torch_dtype is None or torch_dtype == "auto":
        return
    if isinstance(torch_dtype, str):
        torch_dtype = getattr(torch, torch_dtype.replace("torch.", ""))
    name = {torch.float16: "fp16", torch.bfloat16: "bf16", torch.float32: "fp32"}.get(torch_dtype)
    if name is None:
        raise ValueError(f"{who}: unsupported dtype (float16, bfloat16 or float32)")
    if name != "fp32":
        runtime.request_half(torch_dtype, who)
    config.valley_precision = name


def build_vision_tower(config_or_name=None, device="cuda:0", state_dict: Optional[Dict] = None, precision: Optional[str] = None,
                       **kw) -> HipCLIPVisionTower:
    """Factory named by the north star (absent in the reference snapshot).  Accepts a
    ``VisionConfig``/HF ``CLIPVisionConfig``-like object, a dict, or a checkpoint directory holding
    ``config.json`` + safetensors/bin weights (the role of ``CLIPVisionModel.from_pretrained`` at
    valley_model.py:38,66).  ``precision``: "bf16" | "fp32" (default: VALLEY_PRECISION)."""
    cfg = VisionConfig(**kw)
    if isinstance(config_or_name, dict):
        cfg = VisionConfig(**{**config_or_name, **kw})
    elif isinstance(config_or_name, str):
        from .checkpoint import load_clip_checkpoint
        cfg, state_dict = load_clip_checkpoint(config_or_name)
    elif config_or_name is not None:
        fields = ("hidden_size", "num_attention_heads", "intermediate_size", "num_hidden_layers", "image_size",
                  "patch_size", "layer_norm_eps", "hidden_act")
        cfg = VisionConfig(**{f: getattr(config_or_name, f) for f in fields if hasattr(config_or_name, f)})
    cls = PreciseCLIPVisionTower if (precision or default_precision()) == "fp32" else HipCLIPVisionTower
    tower = cls(cfg, device=device)
    if state_dict is not None:
        tower.load_state_dict(state_dict)
    return tower


class HipLinear:
    """nn.Linear stand-in exposing ``weight`` / ``bias`` (mm_projector, valley_model.py:54-55)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        self.weight, self.bias = weight, bias
        self.in_features, self.out_features = weight.shape[1], weight.shape[0]

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        shp = x.shape
        if self.weight.dtype == torch.float32:               # fp32 "precise" mode
            y = ops_f32.gemm(x.reshape(-1, shp[-1]).to(torch.float32).contiguous(), self.weight, self.bias)
        else:
            y = ops.gemm(x.reshape(-1, shp[-1]).to(runtime.HALF).contiguous(), self.weight, self.bias)
        return y.view(*shp[:-1], self.out_features)


class ValleyLlamaModel:
    """valley_model.py:21-254."""
    config_class = ValleyConfig

    def __init__(self, config, device="cuda:0"):
        self.config = config
        self.device = torch.device(device)
        self.training = False
        self.patch_pooling_method = "mean"                   # :27
        c = config
        self.precision = resolve_precision(config)
        self.wdtype = torch.float32 if self.precision == "fp32" else runtime.HALF      # dtype of GEMM weights / activations
        engine = PreciseLlama if self.precision == "fp32" else HipLlama
        self.llama = engine(c.hidden_size, c.num_attention_heads, c.intermediate_size, c.num_hidden_layers,
                            c.vocab_size, c.rms_norm_eps, getattr(c, "rope_theta", None) or _rope_theta(c),
                            getattr(c, "max_position_embeddings", 2048), device=self.device)
        self.vision_tower: Optional[HipCLIPVisionTower] = None
        self.mm_projector: Optional[HipLinear] = None
        self.pooling_layer = None                            # v2: SimpleNamespace(weight fp32 [256*H], bias fp32 [1])
        self.delta_encoder = None                            # v3: dict of packed TransformerEncoderLayer weights
        if getattr(config, "mm_vision_tower", None):         # :29-38
            self.vision_tower = build_vision_tower(config.mm_vision_tower, device=self.device, precision=self.precision)
        if getattr(config, "use_patch_importance_pooling", False):   # :40-43
            self.patch_pooling_method = "temporal_importance"
        if getattr(config, "use_delta_transformer", False):  # :45-52
            self.patch_pooling_method = "temporal_transformer"

    @property
    def embed_tokens(self):
        return SimpleNamespace(weight=self.llama.embed)

    def initialize_vision_modules(self, vision_tower, mm_vision_select_layer, pretrain_mm_mlp_adapter=None,
                                  use_patch_importance_pooling=False, use_delta_transformer=False):
        """valley_model.py:59-103."""
        self.config.mm_vision_tower = vision_tower
        if self.vision_tower is None:
            self.vision_tower = vision_tower if isinstance(vision_tower, (HipCLIPVisionTower, PreciseCLIPVisionTower)) \
                else build_vision_tower(vision_tower, device=self.device, precision=self.precision)
        vc = self.vision_tower.config
        num_patches = (vc.image_size // vc.patch_size) ** 2
        self.config.use_mm_proj = True
        self.config.use_patch_importance_pooling = use_patch_importance_pooling
        self.config.use_delta_transformer = use_delta_transformer
        self.config.mm_hidden_size = vc.hidden_size
        self.config.mm_vision_select_layer = mm_vision_select_layer
        if use_patch_importance_pooling:
            self.patch_pooling_method = "temporal_importance"
        if use_delta_transformer:
            self.patch_pooling_method = "temporal_transformer"
        if self.mm_projector is None:
            g = torch.Generator(device=self.device).manual_seed(0)
            w = torch.randn((self.config.hidden_size, vc.hidden_size), generator=g, device=self.device) * vc.hidden_size ** -0.5
            self.mm_projector = HipLinear(w.to(self.wdtype), torch.zeros(self.config.hidden_size, device=self.device))
        if pretrain_mm_mlp_adapter is not None:
            sd = torch.load(pretrain_mm_mlp_adapter, map_location="cpu")
            sd = {k.split(".")[-1]: v for k, v in sd.items()}
            self.mm_projector = HipLinear(sd["weight"].to(self.device, self.wdtype).contiguous(),
                                          sd["bias"].to(self.device, torch.float32).contiguous())
        return dict(image_processor=_clip_image_processor(vision_tower, vc), image_token_len=num_patches, vision_config=vc)

    # ---- visual tokens -----------------------------------------------------------------------------
    def encode_clips(self, images) -> (torch.Tensor, List[int]):
        """images: [B,T,3,224,224] tensor or list of [T_i,3,224,224] (valley_model.py:168-184).
        Returns (pooled bf16 [sum(256+T_i), W], frames per clip).  W = 1024 for mean pooling (projection
        comes after), W = H for max pooling (projection first)."""
        with runtime.stream_lock():
            return self._encode_clips_locked(images)

    def _encode_clips_locked(self, images):
        if isinstance(images, torch.Tensor) and images.dim() == 5 and images.is_contiguous():
            # [B, T, 3, 224, 224] (valley_model.py:179-184): the clips ARE one contiguous run of frames — a view, not a 38 MB copy
            Ts = [int(images.shape[1])] * int(images.shape[0])
            frames = images.to(self.device).view(-1, *images.shape[2:])
        else:
            clips = list(images) if isinstance(images, (list, tuple)) else [images[b] for b in range(len(images))]
            Ts = [int(c.shape[0]) for c in clips]
            frames = torch.cat([c.to(self.device) for c in clips], 0) if len(clips) > 1 else clips[0].to(self.device)
        sel = getattr(self.config, "mm_vision_select_layer", -1)
        feats = self.vision_tower.encode(frames, select_layer=sel)            # fp32 [F,257,1024]
        method = self.patch_pooling_method
        if method not in ("mean", "max", "temporal_importance", "temporal_transformer"):
            raise ValueError(f"unknown patch_pooling_method {method!r}")
        if self.precision == "fp32":
            return self._pool_precise(feats, Ts, method)
        W = 1024
        if method in ("max", "temporal_importance", "temporal_transformer"):
            # max does not commute with the projector: project every token first (reference order)
            x16 = ops.cast_bf16(feats.view(-1, 1024))
            feats = ops.gemm(x16, self.mm_projector.weight, self.mm_projector.bias, out_dtype=torch.float32)
            W = feats.shape[-1]
            feats = feats.view(-1, 257, W)
        if method == "temporal_transformer":
            outs, f0 = [], 0
            for T in Ts:                                             # one encoder pass per clip length
                outs.append(self.temporal_transformer_delta(feats[f0:f0 + T].reshape(-1, W), 1, T).view(-1, W))
                f0 += T
            return (outs[0] if len(outs) == 1 else torch.cat(outs, 0)), Ts
        mode = {"mean": ops.POOL_MEAN, "max": ops.POOL_MAX, "temporal_importance": ops.POOL_IMPORTANCE}[method]
        scores = None
        if method == "temporal_importance":                              # valley_model.py:113-121
            if self.pooling_layer is None:
                raise RuntimeError("temporal_importance pooling needs model.pooling_layer weights")
            scores = ops.temporal_scores(feats.reshape(-1, W), self.pooling_layer.weight, self.pooling_layer.bias, sum(Ts))
        outs, f0 = [], 0
        if len(set(Ts)) == 1:
            pooled = ops.pool_tokens(feats.reshape(-1, W), len(Ts), Ts[0], mode, scores).view(-1, W)
        else:
            for T in Ts:
                sc = None if scores is None else scores[f0:f0 + T].contiguous()
                outs.append(ops.pool_tokens(feats[f0:f0 + T].reshape(-1, W), 1, T, mode, sc).view(-1, W))
                f0 += T
            pooled = torch.cat(outs, 0)
        return pooled, Ts

    def _pool_precise(self, feats: torch.Tensor, Ts: List[int], method: str):
        """fp32 mode, in the REFERENCE's order for every variant: project all tokens (valley_model.py:190), then pool
        (:206-215) — no pool-before-project reordering, no bf16 anywhere.  Returns projected tokens fp32 [sum(256+T), H]."""
        proj = ops_f32.gemm(feats.view(-1, 1024), self.mm_projector.weight, self.mm_projector.bias)
        W = proj.shape[-1]
        if method == "temporal_transformer":                 # round 4: the v3 encoder layer with fp32 operands throughout
            outs, f0 = [], 0
            for T in Ts:
                outs.append(self._temporal_transformer_delta_f32(proj.view(-1, 257, W)[f0:f0 + T].reshape(-1, W), 1, T).view(-1, W))
                f0 += T
            return (outs[0] if len(outs) == 1 else torch.cat(outs, 0)), Ts
        mode = {"mean": ops.POOL_MEAN, "max": ops.POOL_MAX, "temporal_importance": ops.POOL_IMPORTANCE}[method]
        scores = None
        if method == "temporal_importance":
            if self.pooling_layer is None:
                raise RuntimeError("temporal_importance pooling needs model.pooling_layer weights")
            scores = ops.temporal_scores(proj, self.pooling_layer.weight, self.pooling_layer.bias, sum(Ts))
        outs, f0 = [], 0
        for T in Ts:
            sc = None if scores is None else scores[f0:f0 + T].contiguous()
            outs.append(ops_f32.pool_tokens(proj.view(-1, 257, W)[f0:f0 + T].reshape(-1, W), 1, T, mode, sc).view(-1, W))
            f0 += T
        return (outs[0] if len(outs) == 1 else torch.cat(outs, 0)), Ts

    def temporal_transformer_delta(self, feats: torch.Tensor, B: int, T: int) -> torch.Tensor:
        """valley_model.py:123-133 on B clips of T frames (projected feats fp32 [B*T*257, H]) ->
        bf16 [B, 256+T, H].  Only the last time step of the encoder output is consumed (:130), so the
        query projection, out_proj, both LayerNorms and the FFN run on 256 rows per clip."""
        de = self.delta_encoder
        if de is None:
            raise RuntimeError("temporal_transformer pooling needs model.transformer_delta_encoder weights")
        x_all, x16, x32, mean = ops.delta_prep(feats, de["pos"], B, T)
        kv = ops.gemm(x_all, de["w_kv"], de["b_kv"])                       # [B*256*T, 2H]
        q = ops.gemm(x16, de["w_q"], de["b_q"])                            # [B*256, H]
        att = ops.delta_attention(q, kv, T, 8)
        h1 = ops.gemm(att, de["w_o"], de["b_o"], residual=x32, out_dtype=torch.float32)
        y16, y32 = ops.layernorm(h1, de["n1_g"], de["n1_b"], 1e-5, want_f32=True)
        f = ops.gemm(y16, de["w_1"], de["b_1"], epilogue=ops.EPI_RELU)
        h2 = ops.gemm(f, de["w_2"], de["b_2"], residual=y32, out_dtype=torch.float32)
        _, delta = ops.layernorm(h2, de["n2_g"], de["n2_b"], 1e-5, want_f32=True)
        return ops.delta_finish(delta, mean, feats, B, T)

    def _temporal_transformer_delta_f32(self, feats: torch.Tensor, B: int, T: int) -> torch.Tensor:
        """temporal_transformer_delta in the fp32 mode: the same sequence through vly_delta_*_f32 / vly_gemm_f32 / vly_norm_f32
        (valley_model.py:123-133) -> fp32 [B, 256+T, H]."""
        de = self.delta_encoder
        if de is None:
            raise RuntimeError("temporal_transformer pooling needs model.transformer_delta_encoder weights")
        x_all, x32, mean = ops_f32.delta_prep(feats, de["pos"], B, T)
        kv = ops_f32.gemm(x_all, de["w_kv"], de["b_kv"])
        q = ops_f32.gemm(x32, de["w_q"], de["b_q"])
        att = ops_f32.delta_attention(q, kv, T, 8)
        h1 = ops_f32.gemm(att, de["w_o"], de["b_o"], residual=x32)
        y = ops_f32.norm(h1, de["n1_g"], de["n1_b"], 1e-5)
        f = ops_f32.gemm(y, de["w_1"], de["b_1"], epilogue=ops.EPI_RELU)
        h2 = ops_f32.gemm(f, de["w_2"], de["b_2"], residual=y)
        delta = ops_f32.norm(h2, de["n2_g"], de["n2_b"], 1e-5)
        return ops_f32.delta_finish(delta, mean, feats, B, T)

    def project_pooled(self, pooled: torch.Tensor) -> torch.Tensor:
        """pooled bf16 [NV, 1024] -> visual tokens bf16 [NV, H] (mm_projector, valley_model.py:190)."""
        if self.precision == "fp32":
            return pooled                                    # _pool_precise projects before pooling, as the reference does
        if pooled.shape[-1] == self.config.hidden_size and self.patch_pooling_method != "mean":
            return pooled                                    # these variants project every token before pooling
        return ops.gemm(pooled, self.mm_projector.weight, self.mm_projector.bias)

    def embed_inputs(self, input_ids, images=None, visual_tokens: Optional[torch.Tensor] = None,
                     frames_per_clip: Optional[Sequence[int]] = None) -> torch.Tensor:
        """Token embedding + visual splice (valley_model.py:160-247) -> fp32 residual stream [B*S, H]."""
        B, S = input_ids.shape
        ids_host = _ids_to_host(input_ids)
        row_map = ids_host.astype(np.int32).reshape(-1)
        visual = None
        has_vision = self.vision_tower is not None and (S != 1 or self.training)          # :164
        if has_vision and (images is not None or visual_tokens is not None):
            if visual_tokens is None:
                pooled, frames_per_clip = self.encode_clips(images)
                visual = self.project_pooled(pooled)
            else:
                visual = visual_tokens
            row_map = build_row_map(ids_host, frames_per_clip, self.vision_tower.config)
        if row_map.min() < -(0 if visual is None else visual.shape[0]) or row_map.max() >= self.llama.V:
            raise IndexError("index out of range in self")               # torch embedding's error text
        # pinned staging + async copy: a pageable H2D copy is stream-ordered AND blocks the host, i.e. it waits for
        # every kernel queued before it (the whole ViT encode) and the host falls behind the GPU (measured: the
        # host enqueue time of a c2 step was 25.4 of 27.1 ms, 16 ms of it inside this one `.to()`; 5.8 ms now)
        stage = torch.empty(row_map.shape, dtype=torch.int32, pin_memory=True)
        stage.numpy()[...] = row_map
        splice = ops_f32.embed_splice if self.precision == "fp32" else ops.embed_splice
        return splice(stage.to(self.device, non_blocking=True), self.llama.embed, visual)

    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, images=None, return_dict=None,
                visual_tokens: Optional[torch.Tensor] = None, frames_per_clip: Optional[Sequence[int]] = None):
        """valley_model.py:135-254.  ``visual_tokens``/``frames_per_clip`` let a caller that already
        encoded (e.g. the frame-DP path, valley_amd/parallel.py) skip the tower."""
        if inputs_embeds is not None:
            h = inputs_embeds.to(self.device, torch.float32).reshape(-1, self.config.hidden_size).contiguous().clone()
            B, S = inputs_embeds.shape[:2]
        else:
            B, S = input_ids.shape
            h = self.embed_inputs(input_ids, images, visual_tokens, frames_per_clip)
        cache = past_key_values
        if cache is not None and not isinstance(cache, (HipKVCache, F32KVCache)):
            if len(cache) and (not hasattr(cache, "get_seq_length") or cache.get_seq_length()):
                raise TypeError(f"past_key_values must be the HipKVCache a previous forward returned, got {type(cache).__name__}: "
                                "a foreign (HF tuple / DynamicCache) cache cannot be continued on the HIP path")
            cache = None                                       # an EMPTY foreign cache (HF generate's step 0) starts fresh
        if cache is None:
            # a cache nobody sized: prompt + 256 positions (rounded to 128), growing by doubling up to
            # max_position_embeddings when the caller keeps appending (HipKVCache.reserve) — not 2048 deep up front
            limit = max(getattr(self.config, "max_position_embeddings", 2048), S)
            cache = self.llama.new_cache(B, min(limit, (S + 256 + 127) // 128 * 128) if use_cache else S)
            cache.growable, cache.limit = bool(use_cache), limit
        cache.reserve(cache.seq_len + S)
        if attention_mask is not None:
            am = attention_mask.to(self.device)
            if cache.key_valid is None and bool((am == 0).any()):
                cache.key_valid = torch.ones((B, cache.ctx_max), dtype=torch.uint8, device=self.device)
            if cache.key_valid is not None:
                n = min(am.shape[1], cache.ctx_max)
                cache.key_valid[:, :n] = am[:, :n].to(torch.uint8)
        states = [] if output_hidden_states else None      # valley_model.py:282,324-330 -> HF LlamaModel's all_hidden_states
        # valley_model.py:281,324-330 -> HF LlamaModel's all_self_attns: per layer [B, heads, S, kv_len] in the model dtype (HF's
        # eager attention returns the fp32 softmax cast to the query dtype); a separate pass per layer (ops.attention_probs)
        attns = [] if output_attentions else None
        x = self.llama.forward(h, B, S, cache, collect=states, attn=attns)
        return BaseModelOutputWithPast(last_hidden_state=x.view(B, S, -1), past_key_values=cache if use_cache else None,
                                       hidden_states=None if states is None else tuple(t.view(B, S, -1) for t in states),
                                       attentions=None if attns is None else tuple(a.to(self.wdtype) for a in attns))

    __call__ = forward


def _ids_to_host(input_ids: torch.Tensor) -> np.ndarray:
    """Token ids as a host array (the splice's index logic — which rows become visual tokens, the reference's
    ValueErrors — runs on the host, B*S integers).  Host ids (what a tokenizer hands over) cost nothing.  Device ids
    need a D2H copy that is ordered behind whatever is already queued on the stream: free when the forward is the first
    thing queued (the reference's call pattern: ``model(input_ids.cuda(), images=...)`` embeds before it encodes, and so
    does this forward), a stall only for a caller that queued the tower first and then passes device ids — such a
    caller (bench.py's split step) keeps the ids on the host, where they came from."""
    t = input_ids.detach()
    return (t.cpu() if t.is_cuda else t).numpy()


def _clip_image_processor(name_or_tower, vc):
    """valley_model.py:63,99-103 returns ``CLIPImageProcessor.from_pretrained(vision_tower)``.  From a local checkpoint
    directory that is what is loaded; otherwise (no hub access) the processor is built from CLIP's published
    preprocessing constants — short side to ``image_size``, centre crop, CLIP mean/std — which is what the
    ``openai/clip-vit-large-patch14`` preprocessor_config.json holds."""
    try:
        from transformers import CLIPImageProcessor
        if isinstance(name_or_tower, str) and os.path.isdir(name_or_tower) and \
                os.path.exists(os.path.join(name_or_tower, "preprocessor_config.json")):
            return CLIPImageProcessor.from_pretrained(name_or_tower)
        return CLIPImageProcessor(size={"shortest_edge": vc.image_size}, crop_size={"height": vc.image_size, "width": vc.image_size})
    except Exception:  # noqa: BLE001 - transformers build without any image backend
        return None


def _rope_theta(c) -> float:
    rp = getattr(c, "rope_parameters", None)
    if isinstance(rp, dict) and "rope_theta" in rp:
        return float(rp["rope_theta"])
    return 10000.0


class ValleyLlamaForCausalLM:
    """valley_model.py:257-439."""
    config_class = ValleyConfig

    def __init__(self, config, device="cuda:0"):
        from . import lib
        resolve_precision(config)                            # a 16-bit type named by the config picks the library ...
        lib.load()                                           # ... which is loaded here: fail loudly if it is absent
        self.config = config
        self.device = torch.device(device)
        self.model = ValleyLlamaModel(config, device=device)

    # -- module-ish API ------------------------------------------------------------------------------
    def get_model(self):                                     # :269
        return self.model

    def eval(self):
        return self

    def to(self, *args, **kwargs):
        """``model.to(device)`` / ``.to(dtype)`` of the entry points (run_valley.py:39): the engines were built on their
        device in their storage dtype, so a matching request is a no-op — a request for ANOTHER device or ANOTHER dtype
        raises instead of silently answering from what was built."""
        want, dt = kwargs.get("device"), kwargs.get("dtype")
        for a in args:
            if isinstance(a, (str, torch.device, int)):
                want = a
            elif isinstance(a, torch.dtype):
                dt = a
        if want is not None:
            want = torch.device("cuda", want) if isinstance(want, int) else torch.device(want)
            have = self.device
            if want.type != have.type or (want.index is not None and have.index is not None and want.index != have.index):
                raise ValueError(f"this model was built on {have}; build it with device={want!s} instead of moving it "
                                 f"(from_pretrained(..., device=...) / ValleyLlamaForCausalLM(config, device=...))")
        if dt is not None:
            self._require_dtype(dt, f".to({dt})")
        return self

    def _require_dtype(self, dt, who: str):
        have = self.model.wdtype
        if dt == have or not dt.is_floating_point:
            return
        raise ValueError(f"{who}: this model stores its weights and activations in {have} — the storage type is fixed when "
                         f"the model is built and is not converted afterwards.  Ask for it up front: "
                         f"from_pretrained(..., torch_dtype={dt}) / ValleyConfig.valley_precision / VALLEY_PRECISION")

    def half(self):
        """``model.half()`` = ``.to(torch.float16)`` (valley_model.py:430 does it to the frames): a no-op on an fp16 model, an
        error on a bf16 / fp32 one."""
        self._require_dtype(torch.float16, ".half()")
        return self

    def bfloat16(self):
        self._require_dtype(torch.bfloat16, ".bfloat16()")
        return self

    def float(self):
        self._require_dtype(torch.float32, ".float()")
        return self

    @property
    def dtype(self):
        return self.model.wdtype

    @property
    def lm_head(self):
        return SimpleNamespace(weight=self.model.llama.lm_head[:self.model.llama.V])

    def load_state_dict(self, sd: Dict, strict: bool = True):
        """Reference key names: model.*, lm_head.weight, model.mm_projector.{weight,bias},
        model.vision_tower.* (valley/model/apply_delta.py:25,30)."""
        self.model.llama.load_state_dict(sd)
        if "model.mm_projector.weight" in sd:
            self.model.mm_projector = HipLinear(_dev(sd["model.mm_projector.weight"], self.device, self.model.wdtype),
                                                _dev(sd["model.mm_projector.bias"], self.device, torch.float32))
            self.config.use_mm_proj = True
        if "model.pooling_layer.weight" in sd:               # v2 temporal importance (valley_model.py:42)
            self.model.pooling_layer = SimpleNamespace(
                weight=_dev(sd["model.pooling_layer.weight"], self.device, torch.float32).reshape(-1),
                bias=_dev(sd["model.pooling_layer.bias"], self.device, torch.float32).reshape(-1))
        pfx = "model.transformer_delta_encoder.layers.0."
        if pfx + "self_attn.in_proj_weight" in sd:            # v3 temporal transformer (valley_model.py:45-52)
            d, bf, f32 = self.device, self.model.wdtype, torch.float32     # (fp32 mode: fp32 GEMM weights)
            H = self.config.hidden_size
            win, bin_ = _dev(sd[pfx + "self_attn.in_proj_weight"], d, bf), _dev(sd[pfx + "self_attn.in_proj_bias"], d, f32)
            self.model.delta_encoder = dict(
                w_q=win[:H].contiguous(), b_q=bin_[:H].contiguous(), w_kv=win[H:].contiguous(), b_kv=bin_[H:].contiguous(),
                w_o=_dev(sd[pfx + "self_attn.out_proj.weight"], d, bf), b_o=_dev(sd[pfx + "self_attn.out_proj.bias"], d, f32),
                w_1=_dev(sd[pfx + "linear1.weight"], d, bf), b_1=_dev(sd[pfx + "linear1.bias"], d, f32),
                w_2=_dev(sd[pfx + "linear2.weight"], d, bf), b_2=_dev(sd[pfx + "linear2.bias"], d, f32),
                n1_g=_dev(sd[pfx + "norm1.weight"], d, f32), n1_b=_dev(sd[pfx + "norm1.bias"], d, f32),
                n2_g=_dev(sd[pfx + "norm2.weight"], d, f32), n2_b=_dev(sd[pfx + "norm2.bias"], d, f32),
                pos=_dev(sd["model.position_matrix"], d, f32))
        vt = {k[len("model.vision_tower."):]: v for k, v in sd.items() if k.startswith("model.vision_tower.")}
        if vt:
            created = self.model.vision_tower is None
            if created:
                self.model.vision_tower = build_vision_tower(None, device=self.device, precision=self.model.precision)
            # a model checkpoint stores its whole tower (save_pretrained): when this model had to create the tower
            # object itself, from defaults, the stored depth is the tower's depth
            self.model.vision_tower.load_state_dict(vt, shallow_ok=created)
        return self

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, device="cuda:0", **kw):
        """Sharded HF checkpoint directory (config.json + *.safetensors / pytorch_model-*.bin)."""
        from .checkpoint import load_valley_checkpoint
        config, sd = load_valley_checkpoint(path, ValleyConfig)
        config.mm_vision_tower_name = getattr(config, "mm_vision_tower", None)
        apply_torch_dtype(config, torch_dtype, "from_pretrained(torch_dtype=%s)" % torch_dtype)
        return cls.from_state_dict(config, sd, device=device)

    @classmethod
    def from_state_dict(cls, config, sd: Dict, device="cuda:0"):
        """Model from an in-memory state dict with the reference's key names (what from_pretrained, the delta tool and
        the LoRA merge all end in)."""
        tower_name = getattr(config, "mm_vision_tower", None)
        if tower_name is not None and not os.path.isdir(str(tower_name)):
            config.mm_vision_tower = None                    # no hub access: the tower must come from the state dict
        model = cls(config, device=device)
        model.load_state_dict(sd)
        config.mm_vision_tower = tower_name
        return model

    # -- forward ---------------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, return_dict=None, **kw):
        """valley_model.py:272-330.  Returns CausalLMOutputWithPast(logits fp32 [B,S,V], past_key_values)."""
        if labels is not None:
            raise NotImplementedError("training loss (valley_model.py:307-318) is outside the inference hot path")
        # one launch sequence per stream at a time (the reference's worker calls the model from a thread pool,
        # serve/model_worker.py:467-474); requests on different HIP streams run concurrently
        with runtime.stream_lock():
            out = self.model.forward(input_ids=input_ids, attention_mask=attention_mask, past_key_values=past_key_values,
                                     inputs_embeds=inputs_embeds, use_cache=use_cache, output_attentions=output_attentions,
                                     output_hidden_states=output_hidden_states, images=images, **kw)
            hidden = out.last_hidden_state                                    # bf16 [B,S,H] (a workspace view)
            B, S, H = hidden.shape
            logits = self.model.llama.logits(hidden.view(B * S, H)).view(B, S, -1)  # lm_head on ALL positions (:304-305)
        if return_dict is False:
            return (logits, out.past_key_values)
        return CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=out.past_key_values,
                                      hidden_states=getattr(out, "hidden_states", None),
                                      attentions=getattr(out, "attentions", None))

    __call__ = forward

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kwargs):
        """valley_model.py:332-352 (an EMPTY HipKVCache is falsy, so step 0 keeps the whole prompt — the
        behaviour the reference had at its pinned transformers)."""
        if past_key_values:
            input_ids = input_ids[:, -1:]
        if inputs_embeds is not None and past_key_values is None:
            model_inputs = {"inputs_embeds": inputs_embeds}
        else:
            model_inputs = {"input_ids": input_ids}
        model_inputs.update({"past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"),
                             "attention_mask": attention_mask, "images": kwargs.get("images", None)})
        return model_inputs

    @torch.no_grad()
    def generate(self, input_ids, images=None, attention_mask=None, max_new_tokens: int = 64, do_sample: bool = False,
                 temperature: float = 1.0, stopping_criteria=None, eos_token_id=None, use_graph=True, **kw):
        """Prefill + per-token KV decode (the loop of serve/model_worker.py:371-394; the reference's CLI
        path reaches the same through HF ``generate``, valley_model.py:432).  Greedy when not sampling or
        temperature < 1e-4, else temperature softmax + multinomial.  Decode steps run through a
        hipGraph-captured DecodeSession (``use_graph=True``), eagerly through it (False) or through the
        generic forward (None)."""
        input_ids = input_ids.to(self.device)
        B, S = input_ids.shape
        ctx = min(getattr(self.config, "max_position_embeddings", 2048), S + max_new_tokens)
        cache = self.model.llama.new_cache(B, max(ctx, S + 1))
        out = self.forward(input_ids=input_ids, images=images, attention_mask=attention_mask, past_key_values=cache,
                           use_cache=True)
        greedy = not (do_sample and temperature >= 1e-4)

        def pick(last):
            return ops.argmax(last).to(torch.long) if greedy else \
                torch.multinomial(torch.softmax(last / temperature, dim=-1), num_samples=1).view(B)

        eos = None
        if eos_token_id is not None:
            eos = torch.as_tensor([eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id), device=self.device)
        pad = kw.get("pad_token_id", getattr(self.config, "pad_token_id", None))
        if pad is None:
            pad = int(eos[0]) if eos is not None else 0
        finished = torch.zeros((B,), dtype=torch.bool, device=self.device)     # HF: a finished row emits pad from then on
        token = pick(out.logits[:, -1, :].contiguous())
        seq = torch.cat([input_ids, token[:, None]], dim=1)
        if B > 8 or self.model.precision == "fp32":
            use_graph = None                                 # the GEMV decode session is bf16, for <= 8 sequences
        sess = None
        if use_graph is not None:
            from .decode import DecodeSession
            sess = DecodeSession(self.model.llama, cache, use_graph=bool(use_graph))
            sess.begin(token)
        mask = attention_mask
        for _ in range(max_new_tokens - 1):
            if eos is not None:
                finished |= torch.isin(token, eos)
            if stopping_criteria is not None:
                # HF StoppingCriteriaList: generation stops as soon as ANY criterion fires; a criterion may answer
                # per row (bool tensor [B]) or for the whole batch (python bool, as the reference's keyword criterion)
                for c in stopping_criteria:
                    r = c(seq, None)
                    finished |= r.to(self.device).view(-1) if isinstance(r, torch.Tensor) else torch.full_like(finished, bool(r))
            if bool(finished.all()) or cache.seq_len + 1 > cache.ctx_max:
                break
            if sess is not None:
                nxt = sess.step()
                if greedy:
                    token = nxt.to(torch.long).clone()
                else:
                    token = pick(sess.logits[:, :self.model.llama.V])
                if bool(finished.any()) or not greedy:
                    token = torch.where(finished, torch.full_like(token, pad), token)
                    sess.tok.copy_(token.to(torch.int32))
            else:
                if mask is not None:
                    mask = torch.cat([mask.to(self.device), torch.ones((B, 1), dtype=mask.dtype, device=self.device)], dim=1)
                out = self.forward(input_ids=token[:, None], attention_mask=mask, past_key_values=cache, use_cache=True)
                token = torch.where(finished, torch.full_like(token, pad), pick(out.logits[:, -1, :].contiguous()))
            seq = torch.cat([seq, token[:, None]], dim=1)
        ops.sk_poll_async(self.device)
        torch.cuda.current_stream().synchronize()            # the caller decodes the tokens next; a stream-K hand-off
        ops.sk_check_polled(self.device)                     # failure anywhere in this generation is reported here at the latest
        if sess is not None:
            sess.check()                                     # ticket counters / grid-barrier abort word of the decode launches
        return seq

    # -- tokenizer / prompt glue -----------------------------------------------------------------------
    def resize_token_embeddings(self, n: int):
        ll = self.model.llama
        if n == ll.V:
            return
        H, d = ll.H, self.device
        emb = torch.zeros((n, H), dtype=ll.embed.dtype, device=d)
        keep = min(n, ll.V)
        emb[:keep] = ll.embed[:keep]
        head = torch.zeros(((n + 7) // 8 * 8, H), dtype=ll.lm_head.dtype, device=d)
        head[:keep] = ll.lm_head[:keep]
        ll.embed, ll.lm_head, ll.V, ll.Vpad = emb, head, n, head.shape[0]
        self.config.vocab_size = n

    def initialize_vision_tokenizer(self, tokenizer):
        """valley_model.py:354-379."""
        vc = self.get_model().vision_tower.config
        vc.use_im_start_end = True
        tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_VIDEO_FRAME_TOKEN], special_tokens=True)
        self.resize_token_embeddings(len(tokenizer))
        num_new = tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN, DEFAULT_VI_START_TOKEN,
                                        DEFAULT_VI_END_TOKEN], special_tokens=True)
        self.resize_token_embeddings(len(tokenizer))
        vc.im_start_token, vc.im_end_token = tokenizer.convert_tokens_to_ids([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN])
        vc.vi_start_token, vc.vi_end_token = tokenizer.convert_tokens_to_ids([DEFAULT_VI_START_TOKEN, DEFAULT_VI_END_TOKEN])
        vc.vi_frame_token = tokenizer.convert_tokens_to_ids(DEFAULT_VIDEO_FRAME_TOKEN)
        if num_new > 0:
            ll = self.model.llama
            ll.embed[ll.V - num_new:ll.V] = ll.embed[:ll.V - num_new].float().mean(0, keepdim=True).to(ll.embed.dtype)
            ll.lm_head[ll.V - num_new:ll.V] = ll.lm_head[:ll.V - num_new].float().mean(0, keepdim=True).to(ll.lm_head.dtype)
        vc.im_patch_token = tokenizer.convert_tokens_to_ids([DEFAULT_IMAGE_PATCH_TOKEN])[0]

    def build_inputs(self, tokenizer, messages):
        """valley_model.py:381-402 (including its role spellings and the hard-coded 8 frame tokens)."""
        prompt = ''
        for m in messages:
            if m['role'] == 'system':
                prompt += m['content'] + '\n\n' + '###'
            elif m['role'] == 'user':
                replace_token = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_PATCH_TOKEN * 256 + DEFAULT_IM_END_TOKEN + \
                    DEFAULT_VI_START_TOKEN + DEFAULT_VIDEO_FRAME_TOKEN * 8 + DEFAULT_VI_END_TOKEN
                if '<video>' in m['content'] or '<image>' in m['content']:
                    message = m['content'].replace('<video>', replace_token)
                    message = message.replace('<image>', replace_token)
                    prompt += ' ' + 'Human' + ": " + message + ' \n' + '###'
            elif m['role'] == 'assistent':
                prompt += ' ' + 'Assistent' + ": " + m['content'] + ' \n' + '###'
            else:
                raise ValueError("Role is only suport \"assistent\", \"human\" and \"system\".")
        if DEFAULT_IM_START_TOKEN not in prompt:
            raise ValueError("You need to specify the <video> token in the query")
        tokenizer.padding_side = 'left'
        return tokenizer([prompt], padding=True)

    def process_response(self, outputs):
        """valley_model.py:404-422."""
        output = []
        for out in outputs:
            while True:
                cur_len = len(out)
                out = out.strip()
                for pattern in ['###', 'Assistant:', 'Response:', 'Valley:']:
                    if out.startswith(pattern):
                        out = out[len(pattern):].strip()
                if len(out) == cur_len:
                    break
            try:
                index = out.index('###')
            except ValueError:
                out += '###'
                index = out.index("###")
            output.append(out[:index].strip())
        return output

    @torch.no_grad()
    def completion(self, tokenizer, video, message: list, gen_kwargs: dict, device=None):
        """valley_model.py:424-439.  ``video`` is a path (decoded by valley_amd.video.load_video) or an
        already preprocessed [3,T,224,224] tensor."""
        from .video import KeywordsStoppingCriteria, load_video
        inputs = self.build_inputs(tokenizer, message)
        input_ids = torch.as_tensor(inputs.input_ids).to(self.device)
        if isinstance(video, (np.ndarray, torch.Tensor)) and video.dtype in (np.uint8, torch.uint8):
            from .video import load_video_gpu               # decoded frames [N,H,W,3]: preprocess on the GPU
            images = load_video_gpu(video, device=self.device)
        else:
            images = video if isinstance(video, torch.Tensor) else load_video(video)
            images = images.permute(1, 0, 2, 3).unsqueeze(0)
        stopping = KeywordsStoppingCriteria(['###'], tokenizer, input_ids)
        gk = {k: v for k, v in gen_kwargs.items() if k in ("max_new_tokens", "do_sample", "temperature", "eos_token_id")}
        output_ids = self.generate(input_ids=input_ids, images=images, stopping_criteria=[stopping], **gk)
        n_in = input_ids.shape[1]
        n_diff = (input_ids != output_ids[:, :n_in]).sum().item()
        if n_diff > 0:
            print(f'[Warning] {n_diff} output_ids are not the same as the input_ids')
        outputs = tokenizer.batch_decode(output_ids[:, n_in:], skip_special_tokens=True)
        return self.process_response(outputs)


def _dev(t, device, dtype):
    if not isinstance(t, torch.Tensor):
        t = torch.from_numpy(np.ascontiguousarray(t))
    return t.to(device=device, dtype=dtype).contiguous()


def _register_auto():
    """valley_model.py:441-442: ``AutoConfig.register("valley", ValleyConfig)`` and
    ``AutoModelForCausalLM.register(ValleyConfig, ValleyLlamaForCausalLM)`` so that ``AutoConfig.from_pretrained`` /
    ``AutoModelForCausalLM.from_pretrained`` on a Valley checkpoint resolve to this implementation."""
    if LlamaConfig is None:
        return False
    try:
        from transformers import AutoConfig, AutoModelForCausalLM
        AutoConfig.register("valley", ValleyConfig, exist_ok=True)
        AutoModelForCausalLM.register(ValleyConfig, ValleyLlamaForCausalLM, exist_ok=True)
        return True
    except Exception:  # noqa: BLE001 - a transformers build that refuses non-PreTrainedModel classes
        return False


AUTO_REGISTERED = _register_auto()
