#!/usr/bin/env python3
"""Error statistics of the HIP GEMMs against float64, next to torch's CPU f32 GEMM on the same operands: rms and max of
(C - C64) / rms(C64), and the bias of the error along the sign of the result (a truncating accumulation pulls towards zero)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
from bigc_hip import _lib
lib = _lib.load()
dev = "cuda:0"
g = torch.Generator().manual_seed(3)
u = 2.0 ** -24
for name, M, N, K, relu in [("K=2048 N(0,1) x xavier", 2048, 512, 2048, False), ("K=512 relu-like", 2048, 512, 512, True), ("K=3072 relu-like", 2048, 512, 3072, True)]:
    A = torch.randn(M, K, generator=g)
    if relu: A = A.clamp_min(0)
    W = torch.randn(N, K, generator=g) * (2.0 / (N + K)) ** 0.5
    ref = A.double() @ W.double().t()
    s = float(ref.pow(2).mean().sqrt())
    def stats(C):
        e = (C.double() - ref)
        return "rms %.2f u  max %.1f u  bias along sign(C) %+.3f u" % (float(e.pow(2).mean().sqrt()) / s / u, float(e.abs().max()) / s / u, float((e * ref.sign()).mean()) / s / u)
    print(name)
    print("   torch CPU f32        :", stats(A @ W.t()))
    Ad, Wd = A.to(dev), W.to(dev)
    for mode, label in ((0, "HIP split f16x3"), (1, "HIP exact f32 MFMA")):
        lib.bigc_linear_set_exact(mode)
        C = torch.empty(M, N, device=dev)
        _lib.check(lib.bigc_linear_f32(_lib.current_stream(), Ad.data_ptr(), K, None, 1, K, Wd.data_ptr(), K, None, None, 0, C.data_ptr(), N, M, N, K, 0))
        torch.cuda.synchronize()
        print("   %-21s:" % label, stats(C.cpu()))
    lib.bigc_linear_set_exact(0)
    # gemm_ps: the operand stored as f16 pairs ("hr8"), ONE accumulator for all three MFMAs of a product
    if K % 32 == 0:
        Y = torch.empty((M, 4 * K), dtype=torch.uint8, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.bigc_split_rows_f32(_lib.current_stream(), Ad.data_ptr(), K, Y.data_ptr(), 4 * K, M, K, flag.data_ptr()))
        C = torch.empty(M, N, device=dev)
        _lib.check(lib.bigc_gemm_ps(_lib.current_stream(), Y.data_ptr(), 4 * K, 1, None, 1, Wd.data_ptr(), K, None, None, 0, C.data_ptr(), N, None, 0, M, N, K, 0, flag.data_ptr()))
        torch.cuda.synchronize()
        print("   %-21s:" % "HIP gemm_ps (hr8 A)", stats(C.cpu()), " (vs the f64 product of the UNSPLIT operand)")
    print("   (rms of C = 1 unit; u = 2^-24)")
