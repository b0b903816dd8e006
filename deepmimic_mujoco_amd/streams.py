"""HIP streams that really run side by side.

A HIP stream is bound to one of a few hardware queues; two streams that land on the same queue execute one after the other,
and the sub-batch pipelines built on them (the two halves of a G1 batch, the double-buffered rollout) then run SLOWER than one
batch on one stream.  Measured on MI355X with streams from torch's pool: about one pair in six shares a queue (two G1 half
batches: 4.4 ms per step side by side, 6.7 ms on one queue, 5.0 ms as a single batch).  `concurrent_streams` probes candidates
with two ~0.3 ms spin kernels and keeps only streams that overlap with every stream already chosen.
"""
import time


def _serialized(torch, device, a, b, cycles=600_000):
    def run(streams):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0
    run([a, b])                                   # first use binds the queues
    one = min(run([a]), run([b]), run([a]))
    two = min(run([a, b]), run([a, b]))
    return two > 1.6 * one


_CACHE = {}


def concurrent_streams(device, k, max_candidates=24):
    """k streams on `device` (a torch.device) that pairwise execute concurrently; falls back to whatever it has after
    `max_candidates` tries (the result is then correct but may not overlap).  The set is found once per process and device and
    shared by every caller (a process has few hardware queues: every further stream raises the odds of sharing one; users order
    their work with events, so sharing streams between them is safe)."""
    import torch
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    have = _CACHE.setdefault(key, [])
    if len(have) >= k:
        return have[:k]
    chosen, spare = list(have), []
    for _ in range(max_candidates):
        if len(chosen) == k:
            break
        c = torch.cuda.Stream(device=device)
        if all(not _serialized(torch, device, c, s) for s in chosen):
            chosen.append(c)
        else:
            spare.append(c)
    while len(chosen) < k:
        chosen.append(spare.pop() if spare else torch.cuda.Stream(device=device))
    _CACHE[key] = chosen
    return chosen[:k]
