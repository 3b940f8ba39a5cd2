"""Picking HIP streams that really run concurrently.

ROCm multiplexes HIP streams onto a few hardware queues (4 by default).  Two streams that land on the
same hardware queue serialize completely, and the legacy default stream overlaps poorly with every other
stream (measured on MI355X, scripts/queue_probe.py: two 10.8 ms chains of 64-workgroup GEMMs take
12.3 ms on a good pair, 17.5 ms on default + pool stream, 21.5 ms on a colliding pair).  The training
graph runs its teacher chain, student chain and the two weight-gradient/optimizer chains on four
streams, so they are chosen by measurement once per process: candidates are created, and a stream is
kept only if a short chain on it overlaps with a chain on every stream already kept.
"""
from __future__ import annotations

import time

import torch

from . import ops

_cache = {}


def _chain(bufs, n):
    a, b, c = bufs
    for _ in range(n):
        ops.gemm_nt(a, b, 1024, 1024, 4096, c)          # 64 workgroups x ~45 us: a quarter of the chip


def _timed(streams, bufs, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, bf in zip(streams, bufs):
        with torch.cuda.stream(s):
            _chain(bf, n)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def concurrent_streams(device, count=4, candidates=12, chain=24, verbose=False):
    """Returns ``count`` pool streams (never the default stream) that pairwise overlap.  Falls back to the
    first candidates if the probe cannot find enough (the graph is still correct, only slower)."""
    device = torch.device(device)
    key = (device.index, count)
    if key in _cache:
        return _cache[key]
    with torch.cuda.device(device):
        mk = lambda *s: torch.zeros(*s, device=device, dtype=torch.bfloat16)
        bufs = [(mk(1024, 4096), mk(1024, 4096), torch.empty(1024, 1024, device=device)) for _ in range(2)]
        # EVC_MAIN_PRIORITY=1 (experiment): the first stream - the training graph's main stream, which carries the teacher's
        # chain - is created with high priority, so its workgroups are dispatched ahead of the side streams' waiting ones
        import os
        hi = os.environ.get("EVC_MAIN_PRIORITY") == "1"
        cand = [torch.cuda.Stream(device, priority=-1 if (hi and i == 0) else 0) for i in range(candidates)]
        _timed([cand[0]], bufs, 4)                                        # warm-up (module load, clocks)
        single = min(_timed([cand[0]], bufs, chain) for _ in range(2))
        kept, report = [cand[0]], []
        for s in cand[1:]:
            if len(kept) == count:
                break
            ratios = [min(_timed([k, s], bufs, chain) for _ in range(2)) / single for k in kept]
            report.append([round(r, 2) for r in ratios])
            if max(ratios) < 1.45:
                kept.append(s)
        if verbose:
            print("concurrent_streams: single chain %.2f ms, pair/single ratios per candidate: %s -> kept %d"
                  % (single * 1e3, report, len(kept)))
        for s in cand:                                                    # fallback: fill up with unused candidates
            if len(kept) == count:
                break
            if s not in kept:
                kept.append(s)
    _cache[key] = kept
    return kept


_masked = []        # (keeps the raw streams alive for the life of the process)


def cu_masked_stream(device, cus_per_xcd, first=0, xcds=8, total_cus=256):
    """A stream whose kernels run only on `cus_per_xcd` compute units of every XCD, starting at per-XCD index `first`
    (hipExtStreamCreateWithCUMask through evc_stream_create_cu_mask; mask bit i = CU i // xcds of XCD i % xcds).
    Returns a torch.cuda.ExternalStream."""
    import ctypes as C
    from . import _lib
    device = torch.device(device)
    words = (total_cus + 31) // 32
    mask = (C.c_uint32 * words)()
    for k in range(first, first + cus_per_xcd):
        for x in range(xcds):
            bit = k * xcds + x
            if bit < total_cus:
                mask[bit // 32] |= 1 << (bit % 32)
    out = C.c_void_p()
    with torch.cuda.device(device):
        _lib.call("evc_stream_create_cu_mask", C.cast(mask, C.c_void_p), words, C.cast(C.pointer(out), C.c_void_p))
    st = torch.cuda.ExternalStream(out.value, device=device)
    _masked.append((out.value, st))
    return st
