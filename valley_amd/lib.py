"""ctypes binding of libvalley_hip.so (include/valley_hip.h).

This is the ONLY compute backend of the package: if the library is missing or does not export the
expected ABI the import fails loudly — there is no PyTorch / CPU fallback path."""
from __future__ import annotations

import ctypes
import os
import threading

import torch
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_uint, c_void_p

from . import build as _build

_LIB = None
_LOAD_LOCK = threading.Lock()

_P = c_void_p
_SIGS = {
    "vly_abi_version": (c_int, []),
    "vly_last_error": (c_char_p, []),
    "vly_storage_dtype": (c_int, []),
    "vly_gemm_bf16": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_gemm_tile_for": (c_int, [c_int, c_int]),
    "vly_gemm_bf16_streamk": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      _P, c_size_t, c_uint, _P]),
    "vly_gemm_streamk_workspace_bytes": (c_size_t, []),
    "vly_gemm_streamk_tile_for": (c_int, [c_int, c_int, c_int]),
    "vly_layernorm": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "vly_rmsnorm": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P]),
    "vly_add_layernorm": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "vly_add_rmsnorm": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "vly_patchify": (c_int, [_P, _P, c_int, _P]),
    "vly_vit_embed_ln": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_float, _P]),
    "vly_vit_attention": (c_int, [_P, _P, c_int, _P]),
    "vly_pool_tokens": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "vly_temporal_scores": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "vly_delta_prep": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vly_delta_attention": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vly_delta_finish": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vly_delta_prep_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vly_delta_attention_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vly_delta_finish_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vly_embed_splice": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "vly_rope_kv": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "vly_llama_attention": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "vly_resize_h_u8": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_resize_v_norm": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_incr_i32": (c_int, [_P, c_int, c_int, _P]),
    "vly_gemv_bf16": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_decode_attention_merged": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "vly_llama_attention_probs": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_gemv_rmsnorm_bf16": (c_int, [_P, _P, c_float, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_decode_attention": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P]),
    "vly_decode_attention_rows": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, _P, c_int, _P]),
    "vly_gemm_bf16_splitk2": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_pack_weight_bf16": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "vly_add2_rmsnorm": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "vly_add2_layernorm": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "vly_argmax": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "vly_cast_f32_bf16": (c_int, [_P, _P, c_long, _P]),
    "vly_gemm_bf16_qkv_rope": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_int, _P]),
    "vly_gemm_skinny_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    # fp32 "precise" path
    "vly_norm_split3_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P]),
    "vly_split3_f32": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_gemm_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_attention_f32": (c_int, [_P, c_long, c_int, _P, _P, c_long, c_long, c_int, _P, c_int, _P, c_long, c_int, c_int, c_int,
                                  c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_norm_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_int, _P]),
    "vly_rope_kv_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_patchify_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "vly_pool_tokens_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "vly_embed_splice_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
}
EXPORTS = tuple(_SIGS)
ABI_VERSION = 7
# include/valley_hip.h "EXPERIMENTAL entry points": exported by libvalley_hip_exp.so only (VALLEY_EXPERIMENTAL=1); bound when present
_SIGS_EXPERIMENTAL = {
    "vly_decode_attention_split": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "vly_gemv_attnmerge_bf16": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vly_decode_layers_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "vly_decode_layers": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P, _P]),
}
EXPERIMENTAL = os.environ.get("VALLEY_EXPERIMENTAL", "0") not in ("", "0")


class ValleyHipError(RuntimeError):
    pass


def lib_path() -> str:
    """libvalley_hip.so (bf16 storage) or libvalley_hip_f16.so (VALLEY_PRECISION=fp16); VALLEY_HIP_LIB overrides."""
    from . import runtime
    if EXPERIMENTAL and "VALLEY_HIP_LIB" not in os.environ:
        return _build.LIB_EXP_F16 if runtime.PRECISION == "fp16" else _build.LIB_EXP
    return os.environ.get("VALLEY_HIP_LIB", _build.LIB_F16 if runtime.PRECISION == "fp16" else _build.LIB)


def experimental() -> bool:
    """True when the loaded library carries the experimental entry points (libvalley_hip_exp.so)."""
    return hasattr(load(), "vly_decode_layers") and getattr(load(), "_vly_experimental", False)


def load():
    """Load (once) and type the shared library.  Raises if it is absent or stale."""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LOAD_LOCK:
        return _load_locked()


def _load_locked():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ValleyHipError(
            f"{path} not found: build it with `python -m valley_amd.build` (hipcc --offload-arch=gfx950). "
            "valley_amd has no non-HIP compute path.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ValleyHipError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    lib._vly_experimental = False
    try:
        for name, (res, args) in _SIGS_EXPERIMENTAL.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        lib._vly_experimental = True
    except AttributeError:
        pass
    if lib.vly_abi_version() != ABI_VERSION:
        raise ValleyHipError(f"ABI mismatch: library {lib.vly_abi_version()} vs binding {ABI_VERSION}")
    from . import runtime
    want = 1 if runtime.HALF == torch.float16 else 0
    if lib.vly_storage_dtype() != want:
        raise ValleyHipError(f"{path} stores {'fp16' if lib.vly_storage_dtype() else 'bf16'} but VALLEY_PRECISION asks for "
                             f"{'fp16' if want else 'bf16'} tensors")
    _LIB = lib
    runtime.bind_half("lib.load(%s)" % os.path.basename(path))        # the 16-bit storage type is final from here on
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().vly_last_error().decode(errors="replace")
        raise ValleyHipError(f"{what} failed (rc={rc}): {msg}")
