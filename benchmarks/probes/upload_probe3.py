#!/usr/bin/env python3
"""Host -> device staging under a host that runs ahead of the GPU: which kind of staging memory stalls?
    python benchmarks/probes/upload_probe3.py MODE   (torch_pinned | small_ring | hip_host | hip_host_noncoherent | device_finegrained)"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
from bigc_hip import _lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "torch_pinned"
dev = torch.device("cuda:0")
lib = _lib.load()
hip = ctypes.CDLL("libamdhip64.so")
a = torch.randn(4096, 4096, device=dev)
src = np.random.randn(300 * 1024 // 4).astype(np.float32)
n = src.nbytes
size = (1 << 20) if mode == "small_ring" else (16 << 20)
if mode in ("torch_pinned", "small_ring"):
    keep = torch.empty((size,), dtype=torch.uint8).pin_memory()
    base = keep.data_ptr()
else:
    p = ctypes.c_void_p()
    if mode == "hip_host":
        rc = hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(size), ctypes.c_uint(0))
    elif mode == "hip_host_noncoherent":
        rc = hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(size), ctypes.c_uint(0x80000000))
    else:
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(size), ctypes.c_uint(0x1))
    assert rc == 0, rc
    base = p.value
host = np.ctypeslib.as_array(ctypes.cast(base, ctypes.POINTER(ctypes.c_uint8)), shape=(size,))
srcb = src.view(np.uint8)
off, slow = 0, []
torch.cuda.synchronize()
t_all = time.perf_counter()
for it in range(150):
    for _ in range(4):
        a @ a
    for k in range(4):
        t0 = time.perf_counter()
        if off + n > size:
            off = 0
        host[off:off + n] = srcb
        t1 = time.perf_counter()
        out = torch.empty((src.size,), dtype=torch.float32, device=dev)
        _lib.check(lib.bigc_copy_mapped(_lib.current_stream(), out.data_ptr(), base + off, n))
        off += (n + 255) & ~255
        t2 = time.perf_counter()
        if t2 - t0 > 1e-3:
            slow.append((it, k, round(1e3 * (t1 - t0), 1), round(1e3 * (t2 - t1), 1)))
torch.cuda.synchronize()
total = time.perf_counter() - t_all
ok = bool(np.array_equal(out.cpu().numpy(), src))
print("%-22s %.2f ms per iteration (GPU work alone ~3.9), last upload intact: %s, slow calls (it, call, host copy ms, launch ms): %s"
      % (mode, 1e3 * total / 150, ok, slow[:12]))
