#!/usr/bin/env python3
"""Micro-benchmark of the GEMM on split operands (csrc/gemm_ps.hip) on the GEMM shapes of one 256-video BIG-C step, next to the
in-loop-split kernel (bigc_linear_f32) on the same shapes.
    python benchmarks/gemm_ps_shapes.py [--reps 10]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
from bigc_hip import _lib  # noqa: E402

# (name, M, N, K, A format: 0 f32 / 1 hr8, residual, gather segments, f32 out, hr8 out, launches per step)
SHAPES = [
    ("fc_feat2enti.0 (A f32)", 579295, 512, 2048, 0, False, 0, False, True, 1),
    ("fc_*2enti.2", 579295, 512, 512, 1, False, 0, False, True, 2),
    ("conv k3s2 (gather)", 519168, 512, 3072, 1, False, 3, True, False, 1),
    ("fc_enti2enco.0", 16384, 512, 2048, 1, False, 0, False, True, 1),
    ("enc in_proj", 16384, 1536, 512, 1, False, 0, True, False, 2),
    ("NT x512x512 hr8", 16384, 512, 512, 1, False, 0, True, False, 11),
    ("dec qk", 49152, 1024, 512, 1, False, 0, True, False, 5),
    ("dec 512x512 hr8->hr8", 49152, 512, 512, 1, False, 0, False, True, 15),
    ("dec 512x512 f32->f32", 49152, 512, 512, 0, False, 0, True, False, 15),
    ("dec 512x512 + res f32 A", 49152, 512, 512, 0, True, 0, True, False, 10),
    ("dec 512x512 + res hr8 A", 49152, 512, 512, 1, True, 0, True, False, 10),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", type=int, default=-1)
    args = ap.parse_args()
    lib = _lib.load()
    dev = "cuda:0"
    tot_f, tot_t, tot_t_old = 0.0, 0.0, 0.0
    for si, (name, M, N, K, afmt, res, nseg, of32, ohr8, count) in enumerate(SHAPES):
        if args.only >= 0 and si != args.only:
            continue
        seg = K // nseg if nseg else K
        src = torch.randn(M, seg, device=dev)
        idx = torch.randint(-1, M, (M, nseg), device=dev, dtype=torch.int32) if nseg else None
        W = torch.randn(N, K, device=dev) * 0.02
        b = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev) if res else None
        C = torch.empty(M, N, device=dev) if of32 else None
        Cs = torch.empty(M, 4 * N, dtype=torch.uint8, device=dev) if ohr8 else None
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        if afmt == 1:
            A = torch.empty(M, 4 * seg, dtype=torch.uint8, device=dev)
            _lib.check(lib.bigc_split_rows_f32(_lib.current_stream(), src.data_ptr(), seg, A.data_ptr(), 4 * seg, M, seg, None))
        else:
            A = src

        def run():
            _lib.check(lib.bigc_gemm_ps(_lib.current_stream(), A.data_ptr(), 4 * seg, afmt, _lib.ptr(idx), max(nseg, 1), W.data_ptr(), K, b.data_ptr(),
                                        _lib.ptr(R), N, _lib.ptr(C), N, _lib.ptr(Cs), 4 * N, M, N, K, 1, flag.data_ptr()))
        Cold = C if C is not None else torch.empty(M, N, device=dev)

        def run_old():
            _lib.check(lib.bigc_linear_f32(_lib.current_stream(), src.data_ptr(), seg, _lib.ptr(idx), max(nseg, 1), seg, W.data_ptr(), K,
                                           b.data_ptr(), _lib.ptr(R), N, Cold.data_ptr(), N, M, N, K, 1))
        ms = []
        for fn in (run, run_old):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / args.reps)
        fl = 2.0 * M * N * K
        print("%-26s M=%7d N=%5d K=%5d  %9.1f us %7.1f TFLOP/s   (in-loop split: %9.1f us %7.1f TFLOP/s)  flag=%d"
              % (name, M, N, K, ms[0] * 1e3, fl / ms[0] / 1e9, ms[1] * 1e3, fl / ms[1] / 1e9, int(flag.item())), flush=True)
        tot_f += fl * count
        tot_t += ms[0] * count
        tot_t_old += ms[1] * count
        del A, W, C, Cs, R, idx, src, Cold
    print("step aggregate: %.1f GFLOP in %.2f ms = %.1f TFLOP/s   (in-loop split: %.2f ms = %.1f TFLOP/s)"
          % (tot_f / 1e9, tot_t, tot_f / tot_t / 1e9, tot_t_old, tot_f / tot_t_old / 1e9))


if __name__ == "__main__":
    main()
