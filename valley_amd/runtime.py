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

# Storage dtype of the half-precision tensors (VALLEY_PRECISION, read once at import): "bf16" (default) and "fp32" (the
# validation engines of valley_amd/precise.py; everything else stays bf16) use libvalley_hip.so, "fp16" — the reference's own
# inference dtype, run_valley.py:39 — uses libvalley_hip_f16.so, the same kernels compiled for IEEE half storage.
PRECISION = os.environ.get("VALLEY_PRECISION", "bf16").lower()
if PRECISION not in ("bf16", "fp16", "fp32"):
    raise ValueError(f"VALLEY_PRECISION must be bf16, fp16 or fp32, got {PRECISION!r}")
HALF = torch.float16 if PRECISION == "fp16" else torch.bfloat16

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
