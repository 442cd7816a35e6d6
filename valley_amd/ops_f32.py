"""Typed wrappers of the fp32 "precise" entry points (include/valley_hip.h, vly_*_f32; valley_amd/csrc/precise_f32.hip).
Same conventions as valley_amd.ops: torch device tensors in, raw pointers + the current HIP stream to the C ABI."""
from __future__ import annotations

from typing import Optional

import torch

import os

from . import lib as _lib
from . import runtime
from .ops import EPI_NONE, EPI_SWIGLU, POOL_MEAN, _chk, _ptr, _stream

F32 = torch.float32

# How the fp32 engines contract (VALLEY_F32_GEMM, or set_gemm_mode):
#   "exact" (default): vly_gemm_f32 — fp32 operands on the exact f32-input MFMA (1/16 of the 16-bit MFMA rate);
#   "x3": SPLIT OPERANDS — every fp32 operand as a 16-bit pair hi + lo, three partial products (hi.hi + hi.lo + lo.hi) as ONE
#         vly_gemm_bf16 over 3 K with fp32 accumulation and fp32 output (include/valley_hip.h: vly_split3_f32).  ~2^-16 relative per
#         term instead of 2^-24: the engine that keeps logits within 1e-3 of the fp32 reference at a third of the production rate
#         (VERDICT r5 #3).  Everything that is not a GEMM (norms, attention, RoPE, softmax, pooling) stays on the fp32 kernels.
GEMM_MODE = os.environ.get("VALLEY_F32_GEMM", "exact").lower()
if GEMM_MODE not in ("exact", "x3"):
    raise ValueError(f"VALLEY_F32_GEMM must be exact or x3, got {GEMM_MODE!r}")
X3_MIN_N = 256          # narrower GEMMs (the v2 pooling score, N = 1) stay on vly_gemm_f32


def set_gemm_mode(mode: str) -> None:
    global GEMM_MODE
    if mode not in ("exact", "x3"):
        raise ValueError(mode)
    GEMM_MODE = mode


class X3Act:
    """The fp32 PRE-activation output of a split-operand GEMM whose activation (quick_gelu | ReLU | SwiGLU) is applied by the next
    GEMM's operand split — one pass over the tensor instead of three.  Deliberately not a tensor: only gemm() may consume it."""

    def __init__(self, pre: torch.Tensor, epilogue: int):
        self.pre, self.epilogue = pre, epilogue
        M, N = pre.shape
        self.shape = (M, N // 2 if epilogue == EPI_SWIGLU else N)

    def materialize(self) -> torch.Tensor:
        """fp32 activated tensor (callers other than gemm(): tests)."""
        x = self.pre
        if self.epilogue == EPI_SWIGLU:
            return x[:, 0::2] / (1 + torch.exp(-x[:, 0::2])) * x[:, 1::2]
        return torch.relu(x) if self.epilogue == 3 else x / (1 + torch.exp(-1.702 * x))


class X3Split:
    """An activation that exists only as its split-operand image [hi | hi | lo] (norm_for_gemm in x3 mode): gemm() alone consumes it."""

    def __init__(self, a3: torch.Tensor, K: int):
        self.a3, self.K = a3, K
        self.shape = (a3.shape[0], K)


_X3_WEIGHTS = {}        # id(fp32 weight tensor) -> (weakref to it, its version, its [hi | lo | hi] image in the tile kernels' block layout)


def _split3(x: torch.Tensor, K: int, epilogue: int, order: int) -> torch.Tensor:
    M = x.shape[0]
    Kp = (K + 63) // 64 * 64
    out = torch.empty((M, 3 * Kp), dtype=runtime.HALF, device=x.device)
    rc = _lib.load().vly_split3_f32(x.data_ptr(), x.stride(0), out.data_ptr(), M, K, Kp, epilogue, order, _stream())
    _lib.check(rc, "vly_split3_f32")
    return out


def _x3_weight(w: torch.Tensor):
    from . import ops
    import weakref
    key = id(w)
    hit = _X3_WEIGHTS.get(key)
    if hit is not None and hit[0]() is w and hit[1] == w._version:
        return hit[2]
    w3 = _split3(w if w.stride(0) % 4 == 0 else w.contiguous(), w.shape[1], EPI_NONE, 1)
    pw = ops.PackedWeight(w3)
    pw.plain = None                                          # (only the block-ordered copy is read: 6 bytes per parameter)
    del w3
    _X3_WEIGHTS[key] = (weakref.ref(w, lambda _r, k=key: _X3_WEIGHTS.pop(k, None)), w._version, pw)   # (dies with the tensor: ids are reused)
    return pw


def _gemm_x3(a, w, bias, residual, epilogue, out):
    from . import ops
    K = w.shape[1]
    if isinstance(a, X3Split):
        assert a.K == K
        a3 = a.a3
    elif isinstance(a, X3Act):
        a3 = _split3(a.pre, K, a.epilogue, 0)
    else:
        a3 = _split3(a, K, EPI_NONE, 0)
    pw = _x3_weight(w)
    M, N = a3.shape[0], w.shape[0]
    if epilogue == EPI_NONE:
        if out is None:
            out = torch.empty((M, N), dtype=F32, device=a3.device)
        ops.gemm_tiles_tuned(a3, pw, bias, residual, out)
        return out
    assert residual is None and out is None
    pre = torch.empty((M, N), dtype=F32, device=a3.device)
    ops.gemm_tiles_tuned(a3, pw, bias, None, pre)
    return X3Act(pre, epilogue)


def gemm(a, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         epilogue: int = EPI_NONE, out: Optional[torch.Tensor] = None):
    """out[M,N'] = epi(a[M,K] @ w[N,K]^T + bias) + residual, fp32 operands: on the exact f32 MFMA, or (GEMM_MODE "x3") as three 16-bit
    partial products — then a call with an activation returns an X3Act that only the next gemm() may consume."""
    if GEMM_MODE == "x3" and w.shape[0] >= X3_MIN_N and w.shape[1] % 4 == 0 and runtime.HALF == torch.bfloat16:
        return _gemm_x3(a, w, bias, residual, epilogue, out)
    if isinstance(a, X3Act):
        a = a.materialize().contiguous()
    if isinstance(a, X3Split):
        raise TypeError("an X3Split operand reached the exact GEMM: norm_for_gemm() and gemm() must run under the same GEMM_MODE")
    _chk(a, F32, "a", contiguous=False)
    _chk(w, F32, "w", contiguous=False)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1, (a.shape, w.shape)
    No = N // 2 if epilogue == EPI_SWIGLU else N
    if out is None:
        out = torch.empty((M, No), dtype=F32, device=a.device)
    else:
        _chk(out, F32, "out", contiguous=False)
        assert tuple(out.shape) == (M, No) and out.stride(1) == 1
    if bias is not None:
        _chk(bias, F32, "bias")
    if residual is not None:
        _chk(residual, F32, "residual", contiguous=False)
        assert tuple(residual.shape) == (M, N)
    rc = _lib.load().vly_gemm_f32(a.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), M, N, K, a.stride(0),
                                  w.stride(0), out.stride(0), residual.stride(0) if residual is not None else 0, epilogue, _stream())
    _lib.check(rc, "vly_gemm_f32")
    return out


def norm(x: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor], eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm (beta given) or RMSNorm (beta None), fp32 -> fp32."""
    _chk(x, F32, "x")
    M, D = x.shape
    y = out if out is not None else torch.empty_like(x)
    rc = _lib.load().vly_norm_f32(x.data_ptr(), gamma.data_ptr(), _ptr(beta), y.data_ptr(), M, D, eps, 0 if beta is not None else 1,
                                  _stream())
    _lib.check(rc, "vly_norm_f32")
    return y


def norm_for_gemm(x: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor], eps: float, n_out: int):
    """norm() whose ONLY consumer is the next gemm() (n_out = that GEMM's output width): a tensor in the exact mode; in x3 mode the
    split-operand image written by the norm kernel itself (vly_norm_split3_f32) — the fp32 result never goes to memory."""
    if not (GEMM_MODE == "x3" and n_out >= X3_MIN_N and x.shape[1] % 4 == 0 and runtime.HALF == torch.bfloat16):
        return norm(x, gamma, beta, eps)
    _chk(x, F32, "x")
    M, D = x.shape
    Kp = (D + 63) // 64 * 64
    out = torch.empty((M, 3 * Kp), dtype=runtime.HALF, device=x.device)
    rc = _lib.load().vly_norm_split3_f32(x.data_ptr(), gamma.data_ptr(), _ptr(beta), out.data_ptr(), M, D, Kp, eps, 0 if beta is not None else 1,
                                         _stream())
    _lib.check(rc, "vly_norm_split3_f32")
    return X3Split(out, D)


def vit_attention(qkv: torch.Tensor, F: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qkv fp32 [F*257, 3072] (q | k | v, 16 heads x 64) -> fp32 [F*257, 1024]; no mask."""
    _chk(qkv, F32, "qkv")
    assert tuple(qkv.shape) == (F * 257, 3072)
    if out is None:
        out = torch.empty((F * 257, 1024), dtype=F32, device=qkv.device)
    base = qkv.data_ptr()
    rc = _lib.load().vly_attention_f32(base, 257 * 3072, 3072, base + 1024 * 4, base + 2048 * 4, 257 * 3072, 64, 3072, None, 0,
                                       out.data_ptr(), 257 * 1024, 1024, F, 16, 257, 257, 64, 0, 0, _stream())
    _lib.check(rc, "vly_attention_f32")
    return out


def llama_attention(qkv: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, key_valid: Optional[torch.Tensor], B: int, S: int,
                    heads: int, past_len: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Rotated q rows fp32 [B*S, 3*heads*128] + fp32 caches [B,heads,ctx_max,128] -> fp32 [B*S, heads*128]; causal + key validity."""
    _chk(qkv, F32, "qkv")
    _chk(kcache, F32, "kcache")
    _chk(vcache, F32, "vcache")
    H = heads * 128
    ctx_max = kcache.shape[2]
    assert tuple(kcache.shape) == (B, heads, ctx_max, 128) and past_len + S <= ctx_max
    kvs = 0
    if key_valid is not None:
        _chk(key_valid, torch.uint8, "key_valid")
        kvs = key_valid.stride(0)
    if out is None:
        out = torch.empty((B * S, H), dtype=F32, device=qkv.device)
    rc = _lib.load().vly_attention_f32(qkv.data_ptr(), S * 3 * H, 3 * H, kcache.data_ptr(), vcache.data_ptr(), heads * ctx_max * 128,
                                       ctx_max * 128, 128, _ptr(key_valid), kvs, out.data_ptr(), S * H, H, B, heads, S, past_len + S, 128,
                                       1, past_len, _stream())
    _lib.check(rc, "vly_attention_f32")
    return out


def rope_kv(qkv: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, B: int, S: int,
            heads: int, past_len: int):
    _chk(qkv, F32, "qkv")
    _chk(kcache, F32, "kcache")
    _chk(vcache, F32, "vcache")
    ctx_max = kcache.shape[2]
    assert cos.shape[0] >= past_len + S and cos.shape[1] == 64
    rc = _lib.load().vly_rope_kv_f32(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), cos.data_ptr(), sin.data_ptr(), B, S, heads,
                                     past_len, ctx_max, _stream())
    _lib.check(rc, "vly_rope_kv_f32")


def patchify(images: torch.Tensor, k_padded: int = 592) -> torch.Tensor:
    _chk(images, F32, "images")
    F = images.shape[0]
    assert tuple(images.shape[1:]) == (3, 224, 224)
    out = torch.empty((F * 256, k_padded), dtype=F32, device=images.device)
    rc = _lib.load().vly_patchify_f32(images.data_ptr(), out.data_ptr(), F, k_padded, _stream())
    _lib.check(rc, "vly_patchify_f32")
    return out


def pool_tokens(feats: torch.Tensor, B: int, T: int, mode: int = POOL_MEAN, scores: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(feats, F32, "feats")
    W = feats.shape[-1]
    assert feats.numel() == B * T * 257 * W
    out = torch.empty((B, 256 + T, W), dtype=F32, device=feats.device)
    rc = _lib.load().vly_pool_tokens_f32(feats.data_ptr(), out.data_ptr(), B, T, W, mode, _ptr(scores), _stream())
    _lib.check(rc, "vly_pool_tokens_f32")
    return out


def embed_splice(row_map: torch.Tensor, embed: torch.Tensor, visual: Optional[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(row_map, torch.int32, "row_map")
    _chk(embed, F32, "embed")
    if visual is not None:
        _chk(visual, F32, "visual")
    R, H = row_map.numel(), embed.shape[1]
    if out is None:
        out = torch.empty((R, H), dtype=F32, device=embed.device)
    rc = _lib.load().vly_embed_splice_f32(row_map.data_ptr(), embed.data_ptr(), _ptr(visual), out.data_ptr(), R, H, _stream())
    _lib.check(rc, "vly_embed_splice_f32")
    return out


def delta_prep(feats: torch.Tensor, pos: torch.Tensor, B: int, T: int):
    """vly_delta_prep_f32: projected feats fp32 [B*T*257, H] -> (x_all fp32 [B*256*T, H], x_last fp32, mean fp32 [B*256, H])."""
    _chk(feats, F32, "feats")
    _chk(pos, F32, "pos")
    H, d = feats.shape[-1], feats.device
    assert feats.numel() == B * T * 257 * H and pos.shape[0] >= T and pos.shape[1] == H
    x_all = torch.empty((B * 256 * T, H), dtype=F32, device=d)
    x_last = torch.empty((B * 256, H), dtype=F32, device=d)
    mean = torch.empty((B * 256, H), dtype=F32, device=d)
    rc = _lib.load().vly_delta_prep_f32(feats.data_ptr(), pos.data_ptr(), x_all.data_ptr(), x_last.data_ptr(), mean.data_ptr(), B, T, H,
                                        _stream())
    _lib.check(rc, "vly_delta_prep_f32")
    return x_all, x_last, mean


def delta_attention(q: torch.Tensor, kv: torch.Tensor, T: int, nhead: int) -> torch.Tensor:
    _chk(q, F32, "q")
    _chk(kv, F32, "kv")
    nseq, H = q.shape
    assert tuple(kv.shape) == (nseq * T, 2 * H)
    out = torch.empty_like(q)
    rc = _lib.load().vly_delta_attention_f32(q.data_ptr(), kv.data_ptr(), out.data_ptr(), nseq, T, H, nhead, _stream())
    _lib.check(rc, "vly_delta_attention_f32")
    return out


def delta_finish(delta: torch.Tensor, mean: torch.Tensor, feats: torch.Tensor, B: int, T: int) -> torch.Tensor:
    _chk(delta, F32, "delta")
    _chk(mean, F32, "mean")
    H = delta.shape[-1]
    out = torch.empty((B, 256 + T, H), dtype=F32, device=delta.device)
    rc = _lib.load().vly_delta_finish_f32(delta.data_ptr(), mean.data_ptr(), feats.data_ptr(), out.data_ptr(), B, T, H, _stream())
    _lib.check(rc, "vly_delta_finish_f32")
    return out
