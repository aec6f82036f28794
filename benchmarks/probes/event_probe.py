#!/usr/bin/env python3
"""What does waiting on a LONG-COMPLETED event cost while newer work is queued behind it?  (torch.cuda.Event.synchronize / .query
on this ROCm stack.)  python benchmarks/probes/event_probe.py"""
import time

import torch

dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev)


def busy(n):
    for _ in range(n):
        a @ a


busy(4)
torch.cuda.synchronize()
for timing in (False, True):
    for queued in (0, 8, 40):
        ev = torch.cuda.Event(enable_timing=timing)
        busy(1)
        ev.record()
        torch.cuda.synchronize()              # the event is complete
        busy(queued)                          # newer work behind it (~0.9 ms per product)
        t0 = time.perf_counter()
        q = ev.query()
        t1 = time.perf_counter()
        ev.synchronize()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        print("enable_timing=%s, %2d products queued after the event: query() -> %s in %.3f ms, synchronize() %.3f ms"
              % (timing, queued, q, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
# an event recorded while the GPU is busy, waited for after it completed but with newer work queued
for queued in (8, 40):
    busy(2)
    ev = torch.cuda.Event()
    ev.record()
    busy(queued)
    time.sleep(0.005)                         # the two products before the event are done by now (1.8 ms)
    t0 = time.perf_counter()
    q = ev.query()
    t1 = time.perf_counter()
    ev.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    print("recorded mid-stream, %2d products queued after: query() -> %s in %.3f ms, synchronize() %.3f ms" % (queued, q, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
