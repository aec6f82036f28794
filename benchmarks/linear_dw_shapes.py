#!/usr/bin/env python3
"""Micro-benchmark of the weight-gradient GEMM (bigc_linear_dw_f32) on shapes of a 64-video training step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
from bigc_hip import _lib  # noqa: E402

SHAPES = [("fc_feat2enti.0", 144824, 512, 2048), ("fc_*2enti.2", 144824, 512, 512), ("decoder 512x512", 12288, 512, 512),
          ("dec in_proj qk", 12288, 1024, 512), ("enc NT", 4096, 512, 512)]


def main():
    lib = _lib.load()
    dev = "cuda:0"
    only = int(sys.argv[sys.argv.index("--only") + 1]) if "--only" in sys.argv else None      # --only i: shape i, split mode only
    for exact in ((0,) if only is not None else (1, 0)):
        lib.bigc_linear_set_exact(exact)
        print("== exact f32 MFMA" if exact else "== split f16x3")
        for si, (name, M, N, K) in enumerate(SHAPES + [("conv k3s2 (gather)", 129792, 512, 3072)]):
            if only is not None and si != only:
                continue
            gather = "gather" in name
            dY = torch.randn(M, N, device=dev) * 0.1
            X = torch.randn(M, K // 3 if gather else K, device=dev)
            idx = None
            if gather:       # conv taps: rows 2j-1, 2j, 2j+1 of the source (clamped), as the stride-2 conv's index table
                j = torch.arange(M, device=dev)
                idx = torch.stack([2 * j - 1, 2 * j, 2 * j + 1], 1).clamp_(-1, M - 1).to(torch.int32).contiguous()
            dW = torch.zeros(N, K, device=dev)
            db = torch.zeros(N, device=dev)

            def run():
                _lib.check(lib.bigc_linear_dw_f32(_lib.current_stream(), dY.data_ptr(), N, X.data_ptr(), X.shape[1], _lib.ptr(idx), 3 if gather else 1,
                                                  K // 3 if gather else K, dW.data_ptr(), K, db.data_ptr(), M, N, K))
            for _ in range(2):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print("%-18s M=%7d N=%5d K=%5d  %9.1f us  %7.1f TFLOP/s" % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
            if exact == 0 and M <= 12288 and not gather:
                dW.zero_()
                run()
                ref = dY.double().t() @ X.double()
                print("      max err / max|dW| = %.2e" % float((dW.double() - ref).abs().max() / ref.abs().max()))
    lib.bigc_linear_set_exact(0)


if __name__ == "__main__":
    main()
