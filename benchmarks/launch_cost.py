#!/usr/bin/env python3
"""Host cost of one kernel launch through the C ABI (ctypes call + hipLaunchKernelGGL) and the GPU's back-to-back rate for tiny
dependent kernels: python benchmarks/launch_cost.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
from bigc_hip import _lib  # noqa: E402


def main():
    lib = _lib.load()
    x = torch.ones(1024, device="cuda:0")
    y = torch.ones(1024, device="cuda:0")
    s = _lib.current_stream()
    f = lib.bigc_relu_bwd_f32
    for n in (200, 2000):
        for _ in range(20):
            f(s, x.data_ptr(), y.data_ptr(), 1024)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xp, yp = x.data_ptr(), y.data_ptr()
        for _ in range(n):
            f(s, xp, yp, 1024)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%d launches: host enqueue %.2f us each, until the GPU is done %.2f us each" % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
    # the same through torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        x.mul_(1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("torch x.mul_(1): host %.2f us each, until done %.2f us each" % (1e6 * (t1 - t0) / 2000, 1e6 * (t2 - t0) / 2000))


if __name__ == "__main__":
    main()
