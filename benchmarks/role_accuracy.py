#!/usr/bin/env python3
"""Accuracy of the role-attention kernels against a float64 evaluation: `python benchmarks/role_accuracy.py` on a GPU box.
Set BIGC_ROLE_LDS=0 for role_fused_h_kernel, BIGC_LINEAR_EXACT=1 for the f32-MFMA kernel."""
import math, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
from bigc_hip import _lib

lib = _lib.load()
dev = "cuda:0"
g = torch.Generator().manual_seed(1)
B, nq, half, d = 4, 192, 256, 512
ns = [64, 32, 50, 7]
NT = sum(ns)
for gain in (1.0, 4.0):
    p2a = torch.randn(B * nq, 2 * half, generator=g) * gain
    e2a = torch.randn(NT, 2 * half, generator=g)
    enco = torch.randn(NT, d, generator=g)
    off = np.concatenate([[0], np.cumsum(ns)])
    scale = 1.0 / math.sqrt(d)
    att_w, val_w = [], torch.empty(2, B * nq, d, dtype=torch.float64)
    for v in range(B):
        e = e2a[off[v]:off[v + 1]].double(); p = p2a[v * nq:(v + 1) * nq].double()
        s = torch.stack([p[:, :half] @ e[:, :half].t(), p[:, half:] @ e[:, half:].t()], 0) * scale
        a = torch.softmax(s, 2) * torch.softmax(s, 0)
        att_w.append(a.reshape(-1))
        for r in range(2):
            val_w[r, v * nq:(v + 1) * nq] = a[r] @ enco[off[v]:off[v + 1]].double()
    att_w = torch.cat(att_w)
    pd, ed, xd = p2a.to(dev), e2a.to(dev), enco.to(dev)
    offd = torch.from_numpy(off.astype(np.int32)).to(dev)
    att = torch.empty(2 * nq * NT, device=dev); vals = torch.empty(2, B * nq, d, device=dev)
    _lib.check(lib.bigc_role_fused_f32(_lib.current_stream(), pd.data_ptr(), 2 * half, ed.data_ptr(), 2 * half, xd.data_ptr(), d, att.data_ptr(),
                                       vals.data_ptr(), offd.data_ptr(), B, nq, max(ns), half, d, scale))
    torch.cuda.synchronize()
    ea = ((att.cpu().double() - att_w).abs() / att_w.abs().clamp_min(1e-30))
    big = att_w > 1e-3
    print("gain %.0f: att max rel err (entries > 1e-3) %.2e, max abs err %.2e; values max err / max %.2e" %
          (gain, float(ea[big].max()), float((att.cpu().double() - att_w).abs().max()), float((vals.cpu().double() - val_w).abs().max() / val_w.abs().max())))
