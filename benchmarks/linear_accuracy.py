#!/usr/bin/env python3
"""Error of bigc_linear_f32 against an fp64 reference (torch on the GPU), in units of the f32 round-off scale:
err = max |C - C64| / (|A| |W|^T) per element.  Run once with BIGC_LINEAR_EXACT=1 (exact f32 MFMA kernel) and once without
(split-operand f16 kernel) to compare; also exercises the out-of-range fallback."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
from bigc_hip import _lib  # noqa: E402


def run(lib, A, W, b):
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty(M, N, device=A.device)
    _lib.check(lib.bigc_linear_f32(_lib.current_stream(), A.data_ptr(), K, None, 1, K, W.data_ptr(), K, b.data_ptr(), None, N, C.data_ptr(), N, M, N, K, 0))
    return C


def main():
    lib = _lib.load()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(1)
    print("mode:", "exact f32" if os.environ.get("BIGC_LINEAR_EXACT") == "1" else "split f16x3")
    for name, M, N, K, sa, sw in [("N(0,1) x 0.02", 4096, 512, 2048, 1.0, 0.02), ("relu-like x xavier", 4096, 512, 512, 3.0, 0.06),
                                  ("wide range", 2048, 133, 3160, 1.0, 0.02), ("tiny", 1024, 512, 512, 1e-6, 1e-4), ("large", 1024, 512, 512, 3e3, 5.0)]:
        A = torch.randn(M, K, device=dev, generator=g) * sa
        if name == "relu-like x xavier":
            A = A.clamp_min(0)
        if name == "wide range":
            A = A * torch.exp(4 * torch.randn(M, K, device=dev, generator=g))
        W = torch.randn(N, K, device=dev, generator=g) * sw
        b = torch.randn(N, device=dev, generator=g)
        C = run(lib, A, W, b)
        ref = A.double() @ W.double().t() + b.double()
        scale = A.double().abs() @ W.double().abs().t() + b.double().abs()
        err = ((C.double() - ref).abs() / scale)
        rel = (C.double() - ref).abs().max() / ref.abs().max()
        print("%-20s M=%d N=%d K=%d  max err/scale = %.3g (= %.2f x 2^-24)  mean = %.3g  max|err|/max|C| = %.3g"
              % (name, M, N, K, float(err.max()), float(err.max()) * 2 ** 24, float(err.mean()), float(rel)))
    # out-of-range operands fall back to the exact kernel
    A = torch.randn(512, 256, device=dev, generator=g)
    A[7, 3] = 1e6
    A[100, 200] = -1e9
    W = torch.randn(128, 256, device=dev, generator=g)
    b = torch.zeros(128, device=dev)
    C = run(lib, A, W, b)
    ref = (A.double() @ W.double().t()).float()
    print("out-of-range fallback: finite =", bool(torch.isfinite(C).all()), " max rel err = %.3g" % float(((C - ref).abs() / ref.abs().clamp_min(1)).max()))
    A[5, 5] = float("inf")
    C = run(lib, A, W, b)
    print("inf operand: row 5 non-finite =", bool((~torch.isfinite(C[5])).all()), " others finite =", bool(torch.isfinite(C[6:]).all()))


if __name__ == "__main__":
    main()
