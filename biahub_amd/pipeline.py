"""Overlap of the three legs of a plate job on one GPU: upload of unit i + 1, compute of unit i, download of unit i - 1.

The reference's worker (biahub/deskew.py:578-579, deconvolve.py:52-66) uploads a volume, computes and takes the result back,
one after the other; across PCIe that is most of the time of a unit (bench.py ``end_to_end``: 75 ms up, 358 ms compute,
297 ms down for a 512 x 2048 x 2048 uint16 stack whose deskewed float32 result is 17 GB).  The three legs use different
engines — the two DMA directions and the compute units — so a plate's units can be pipelined: three HIP streams, events in
between, pinned host blocks at both ends; nothing here blocks the host until a result is handed out.  ``libbhcore`` runs on
torch's current stream (device.Context.bind_stream), so ``compute`` is any function of this package on device tensors.
"""
from __future__ import annotations

from collections import deque
from typing import Callable, Iterable, Iterator

import torch

from .device import resolve_device, volume_pool


_STREAMS: dict = {}


def _streams(dev):
    key = dev.index
    if key not in _STREAMS:
        _STREAMS[key] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1))
    return _STREAMS[key]


def run_overlapped(items: Iterable, upload: Callable, compute: Callable, download: Callable, device="cuda",
                   depth: int = 2, timeline: list | None = None) -> Iterator:
    """Yield ``download(compute(upload(item)))`` for every item, in order, with the legs of neighbouring items overlapped.

    ``upload(item)`` returns device tensor(s) and must only enqueue work (``tensor.to(dev, non_blocking=True)`` from pinned
    memory); ``compute(uploaded)`` returns device tensor(s); ``download(result)`` enqueues the copy back
    (``pinned.copy_(t, non_blocking=True)``) and returns whatever the caller wants to receive — it is handed out only after
    the copy has finished.  ``depth`` bounds the units in flight per leg (device memory: ``depth`` inputs and results).
    Each callable runs with its own stream current; tensors crossing from one leg to the next are ordered by events and
    kept alive for the consuming stream.  ``timeline`` (a list) receives one ``[h2d0, h2d1, c0, c1, d2h0, d2h1]`` row of
    timing events per unit — ``timeline_ms`` turns them into milliseconds once everything has finished.
    """
    dev = resolve_device(device)
    # The download leg is NOT a DMA transfer on this platform: ROCclr copies device -> pinned host memory with a blit kernel
    # (`__amd_rocclr_copyBuffer`, profiles/r03_d2h_probe.txt — HSA_ENABLE_SDMA / GPU_BLIT_ENGINE_TYPE do not change that), which
    # shares the CUs with the compute leg for as long as the copy takes.  The compute stream therefore gets the higher priority:
    # its workgroups are dispatched ahead of the copy kernel's.
    # The three streams are made once per device: torch's caching allocator hands a freed block back only to the stream it
    # was allocated on, so new streams per call would find none of the previous call's gigabyte buffers reusable (and the
    # blocks of device.volume_pool cost ~30 ms per GB to build).
    s_in, s_out, s_c = _streams(dev)
    staged: deque = deque()   # (uploaded, event): waiting for compute
    landing: deque = deque()  # (handed, event): waiting for the copy back to finish

    def _record(obj, stream):
        for t in (obj if isinstance(obj, (tuple, list)) else (obj,)):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(stream)

    timed = timeline is not None

    def _mark(stream):
        ev = torch.cuda.Event(enable_timing=timed)
        ev.record(stream)
        return ev

    def _compute_and_download():
        up, ev, row = staged.popleft()
        s_c.wait_event(ev)
        with torch.cuda.stream(s_c):
            _record(up, s_c)
            if timed:
                row.append(_mark(s_c))
            res = compute(up)
            computed = _mark(s_c)
        del up
        s_out.wait_event(computed)
        with torch.cuda.stream(s_out):
            _record(res, s_out)
            if timed:
                row += [computed, _mark(s_out)]
            handed = download(res)
            done = _mark(s_out)
        del res
        if timed:
            row.append(done)
            timeline.append(row)
        landing.append((handed, done))

    def _hand_out():
        handed, ev = landing.popleft()
        ev.synchronize()
        return handed

    with torch.cuda.device(dev):
        # per item: upload(i) is queued BEFORE compute(i - 1) and download(i - 1), so a host-side wait inside `compute`
        # (hipFFT planning, a first-use allocation) never keeps the next upload from starting
        for item in items:
            with torch.cuda.stream(s_in):
                row = [_mark(s_in)] if timed else None
                with volume_pool(dev):  # the uploaded volume gets the library's page layout (device.volume_pool)
                    up = upload(item)
                ev = _mark(s_in)
            if timed:
                row.append(ev)
            staged.append((up, ev, row))
            del up
            if len(staged) > 1:
                _compute_and_download()
            while len(landing) > max(1, depth) - 1:
                yield _hand_out()
        while staged:
            _compute_and_download()
        while landing:
            yield _hand_out()


def timeline_ms(timeline: list) -> list:
    """Rows of ``run_overlapped(..., timeline=rows)`` as milliseconds since the first upload started:
    ``[h2d_start, h2d_end, compute_start, compute_end, d2h_start, d2h_end]`` per unit (call after the run has finished)."""
    if not timeline:
        return []
    t0 = timeline[0][0]
    return [[round(t0.elapsed_time(ev), 2) for ev in row] for row in timeline]
