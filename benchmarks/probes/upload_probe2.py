#!/usr/bin/env python3
"""Which part of the staging block's half switch is slow?  python benchmarks/probes/upload_probe2.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
from bigc_hip import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
a = torch.randn(4096, 4096, device=dev)
src = torch.from_numpy(np.random.randn(300 * 1024 // 4).astype(np.float32))
buf = torch.empty((16 << 20,), dtype=torch.uint8).pin_memory()
half, n = 8 << 20, src.numel() * 4
mode = sys.argv[1] if len(sys.argv) > 1 else "event"
events = [None, None]
cur, off = 0, 0
slow = []
T = lambda: time.perf_counter()
for it in range(120):
    for _ in range(4):
        a @ a
    for k in range(4):
        t = [T()]
        switched = False
        if off + n > half:
            switched = True
            if mode == "event":
                ev = torch.cuda.Event()
                ev.record()
                events[cur] = ev
            t.append(T())
            cur ^= 1
            off = 0
            if mode == "event" and events[cur] is not None:
                events[cur].synchronize()
            t.append(T())
        view = buf[cur * half + off:cur * half + off + n].view(torch.float32)
        view.copy_(src)
        t.append(T())
        off += n
        out = torch.empty((src.numel(),), dtype=torch.float32, device=dev)
        t.append(T())
        _lib.check(lib.bigc_copy_mapped(_lib.current_stream(), out.data_ptr(), view.data_ptr(), n))
        t.append(T())
        if t[-1] - t[0] > 1e-3:
            slow.append((it, k, switched, [round(1e3 * (y - x), 2) for x, y in zip(t[:-1], t[1:])]))
torch.cuda.synchronize()
print(mode, "slow calls (iteration, call, switched, ms per part [record, sync,] host copy, empty, launch):")
for s in slow:
    print("   ", s)
