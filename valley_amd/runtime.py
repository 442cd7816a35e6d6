"""Host-side execution context: per-stream serialisation and workspace keys.

The reference's worker calls one model object from FastAPI's thread pool (up to 5 concurrent requests,
serve/model_worker.py:467-474).  The HIP path keeps reusable activation workspaces, so two launch
sequences that target the SAME HIP stream must not interleave (the second would overwrite buffers the
first has queued kernels on): `stream_lock()` serialises them.  Sequences on DIFFERENT streams run
concurrently, each with its own workspaces (`stream_key()` is part of every workspace key) — bench.py
uses that to run two half-batches side by side, so the tail of one GEMM is filled by the other's tiles."""
from __future__ import annotations

import threading

import torch

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
