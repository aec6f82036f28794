#!/usr/bin/env python3
"""Waiting for a PENDING event / the stream / a stamp word: how long after the GPU is done does the host resume?
python benchmarks/probes/event_probe2.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
from bigc_hip import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
a = torch.randn(4096, 4096, device=dev)
word = torch.zeros((2,), dtype=torch.int64).pin_memory()
wnp = word.numpy()
seq = [0]


def busy(n):
    for _ in range(n):
        a @ a


busy(8)
torch.cuda.synchronize()
t0 = time.perf_counter()
busy(20)
torch.cuda.synchronize()
per = (time.perf_counter() - t0) / 20
print("one product: %.3f ms" % (1e3 * per))


def wait_event(after):
    ev = torch.cuda.Event()
    busy(10)
    ev.record()
    busy(after)
    t0 = time.perf_counter()
    ev.synchronize()
    return time.perf_counter() - t0


def wait_stamp(after):
    busy(10)
    seq[0] += 1
    _lib.check(lib.bigc_stamp(_lib.current_stream(), word.data_ptr(), seq[0], None))
    busy(after)
    t0 = time.perf_counter()
    while wnp[0] < seq[0]:
        pass
    return time.perf_counter() - t0


def wait_query(after):
    ev = torch.cuda.Event()
    busy(10)
    ev.record()
    busy(after)
    t0 = time.perf_counter()
    while not ev.query():
        pass
    return time.perf_counter() - t0


for name, fn in (("Event.synchronize", wait_event), ("spin on Event.query", wait_query), ("spin on a stamp word", wait_stamp)):
    for after in (0, 30):
        ts = []
        for _ in range(8):
            torch.cuda.synchronize()
            ts.append(fn(after))
        torch.cuda.synchronize()
        print("%-22s 10 products before (%.1f ms), %2d after: waited %s ms" % (name, 1e4 * per, after, " ".join("%.1f" % (1e3 * t) for t in ts)))
