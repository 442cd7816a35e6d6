"""Host-side execution context: per-stream serialisation and workspace keys.

The reference's worker calls one model object from FastAPI's thread pool (up to 5 concurrent requests,
serve/model_worker.py:467-474).  The HIP path keeps reusable activation workspaces, so two launch
sequences that target the SAME HIP stream must not interleave (the second would overwrite buffers the
first has queued kernels on): `stream_lock()` serialises them.  Sequences on DIFFERENT streams run
concurrently, each with its own workspaces (`stream_key()` is part of every workspace key) — bench.py
uses that to run two half-batches side by side, so the tail of one GEMM is filled by the other's tiles."""
from __future__ import annotations

import os
import threading

import torch

# Storage dtype of the half-precision tensors: "bf16" (default) and "fp32" (the validation engines of valley_amd/precise.py;
# everything else stays bf16) use libvalley_hip.so, "fp16" — the reference's own inference dtype, run_valley.py:39 — uses
# libvalley_hip_f16.so, the same kernels compiled for IEEE half storage.  VALLEY_PRECISION pins it for the process; without it
# the type is bf16 until the first caller that names one — ``from_pretrained(torch_dtype=torch.float16)``, ``.half()``,
# ``.to(torch.float16)``, ``ValleyConfig.valley_precision`` — picks it through request_half(), as long as no library has been
# loaded yet.  Afterwards a request for the OTHER 16-bit type raises (one process hosts one library); modules therefore read
# ``runtime.HALF`` at call time, never at import.
_ENV_PRECISION = os.environ.get("VALLEY_PRECISION")
PRECISION = (_ENV_PRECISION or "bf16").lower()
if PRECISION not in ("bf16", "fp16", "fp32"):
    raise ValueError(f"VALLEY_PRECISION must be bf16, fp16 or fp32, got {PRECISION!r}")
HALF = torch.float16 if PRECISION == "fp16" else torch.bfloat16
_HALF_NAMES = {torch.float16: "fp16", torch.bfloat16: "bf16"}
_bound_by = "VALLEY_PRECISION" if _ENV_PRECISION else None      # who fixed the 16-bit storage type (None: still open)


def half_bound() -> bool:
    """True once the 16-bit storage type can no longer change: VALLEY_PRECISION set, a library loaded, or a model built."""
    from . import lib
    return _bound_by is not None or lib._LIB is not None


def bind_half(who: str) -> None:
    """Called by whatever allocates the first 16-bit tensor of a model (engines, lib.load): the type is final from here on."""
    global _bound_by
    if _bound_by is None:
        _bound_by = who


def request_half(dtype, who: str) -> None:
    """A caller asks for 16-bit storage type ``dtype`` (torch.float16 / torch.bfloat16; "fp16" / "bf16" accepted): select the
    matching library if the choice is still open, do nothing if it is what is bound, and RAISE otherwise — the reference's
    ``from_pretrained(..., torch_dtype=torch.float16)`` (run_valley.py:39, serve/model_worker.py:61,79) must not silently run
    in bf16.  torch.float32 / None are not 16-bit requests and are ignored here (fp32 is ValleyConfig.valley_precision)."""
    global PRECISION, HALF, _bound_by
    if isinstance(dtype, str):
        dtype = {"fp16": torch.float16, "float16": torch.float16, "half": torch.float16, "bf16": torch.bfloat16,
                 "bfloat16": torch.bfloat16}.get(dtype.lower())
    if dtype not in _HALF_NAMES:
        return
    if dtype == HALF:
        if PRECISION != "fp32":
            bind_half(who)
        return
    if half_bound() or PRECISION == "fp32":
        from . import lib
        why = _bound_by or ("a loaded library" if lib._LIB is not None else "VALLEY_PRECISION=fp32")
        raise ValueError(f"{who} asks for {_HALF_NAMES[dtype]} storage, but this process is bound to {_HALF_NAMES[HALF]} "
                         f"(by {why}): one process hosts one of libvalley_hip.so / libvalley_hip_f16.so — set "
                         f"VALLEY_PRECISION={_HALF_NAMES[dtype]} or make the request before anything else is built")
    PRECISION, HALF = _HALF_NAMES[dtype], dtype
    _bound_by = who

_locks = {}
_glock = threading.Lock()


def stream_key() -> int:
    return int(torch.cuda.current_stream().cuda_stream) if torch.cuda.is_available() else 0


def stream_lock() -> threading.RLock:
    sid = stream_key()
    with _glock:
        lk = _locks.get(sid)
        if lk is None:
            lk = _locks[sid] = threading.RLock()
    return lk
