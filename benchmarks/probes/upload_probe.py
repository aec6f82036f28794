#!/usr/bin/env python3
"""How long do host -> device uploads of a few hundred KB take while the GPU is busy, by route (blocking pageable copy, asynchronous
copy from the pinned block, the library's copy kernel reading the pinned block)?  python benchmarks/probes/upload_probe.py"""
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
src = np.random.randn(300 * 1024 // 4).astype(np.float32)


def busy():
    for _ in range(4):
        a @ a


def run(name, fn, reps=30):
    torch.cuda.synchronize()
    worst, t0 = 0.0, time.perf_counter()
    for _ in range(reps):
        busy()
        t1 = time.perf_counter()
        for _ in range(4):
            fn()
        worst = max(worst, time.perf_counter() - t1)
    torch.cuda.synchronize()
    print("%-40s %.2f ms per iteration, slowest host-side upload call group %.2f ms" % (name, 1e3 * (time.perf_counter() - t0) / reps, 1e3 * worst))


busy()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    busy()
torch.cuda.synchronize()
print("GPU work alone: %.2f ms per iteration" % (1e3 * (time.perf_counter() - t0) / 30))
pinned = torch.empty((src.size,), dtype=torch.float32).pin_memory()
dst = torch.empty((src.size,), dtype=torch.float32, device=dev)


def by_kernel():
    pinned.numpy()[:] = src
    _lib.check(lib.bigc_copy_mapped(_lib.current_stream(), dst.data_ptr(), pinned.data_ptr(), src.nbytes))


def by_async():
    pinned.numpy()[:] = src
    dst.copy_(pinned, non_blocking=True)


run("blocking pageable .to()", lambda: torch.from_numpy(src).to(dev))
run("async copy from pinned", by_async)
run("copy kernel from pinned", by_kernel)
os.environ["BIGC_UPLOAD_KERNEL"] = "1"
run("_lib.upload (kernel route)", lambda: _lib.upload(src, dev))
os.environ["BIGC_UPLOAD_KERNEL"] = "0"
run("_lib.upload (blocking route)", lambda: _lib.upload(src, dev))
# the kernel's own duration
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    _lib.check(lib.bigc_copy_mapped(_lib.current_stream(), dst.data_ptr(), pinned.data_ptr(), src.nbytes))
e1.record()
torch.cuda.synchronize()
print("copy kernel, 300 KB: %.1f us per launch" % (1e3 * e0.elapsed_time(e1) / 20))
assert np.array_equal(dst.cpu().numpy(), src)

# where in the staging block do the slow calls happen?
os.environ["BIGC_UPLOAD_KERNEL"] = "1"
st = _lib.staging(dev)
slow = []
for it in range(120):
    busy()
    for k in range(4):
        t1 = time.perf_counter()
        _lib.upload(src, dev)
        dt = time.perf_counter() - t1
        if dt > 1e-3:
            slow.append((it, k, st.cur, st.off, round(1e3 * dt, 2)))
torch.cuda.synchronize()
print("slow upload calls (iteration, call, half, offset after, ms):", slow)
