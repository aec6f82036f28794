#!/usr/bin/env python3
"""Micro-benchmark of the linear_f32 kernel on the GEMM shapes of one 256-video BIG-C step.
    python benchmarks/linear_shapes.py [--reps 20]
Prints TFLOP/s per shape and the flops-weighted aggregate (same weighting as bench.py's roofline)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
from bigc_hip import _lib  # noqa: E402

# (name, M, N, K, residual, gather_segments, launches per step)
SHAPES = [
    ("fc_feat2enti.0", 579295, 512, 2048, False, 0, 1),
    ("fc_*2enti.2", 579295, 512, 512, False, 0, 2),
    ("fc_bbox2enti.0", 579295, 512, 8, False, 0, 1),
    ("conv k3s2 (gather)", 519168, 512, 3072, False, 3, 1),
    ("fc_enti2enco.0", 16384, 512, 2048, False, 0, 1),
    ("enc in_proj", 16384, 1536, 512, False, 0, 2),
    ("NT x512x512", 16384, 512, 512, False, 0, 11),
    ("dec qk", 49152, 1024, 512, False, 0, 5),
    ("dec 512x512", 49152, 512, 512, False, 0, 30),
    ("dec 512x512 + res", 49152, 512, 512, True, 0, 20),
    ("head 133x3160", 49152, 133, 3160, True, 0, 1),
]


# the same layers at the training step's 8-video size (bench.py --mode train): 512 tracklet rows, 1536 query rows
TRAIN_SHAPES = [
    ("NT=512 x512x512", 512, 512, 512, False, 0, 30),
    ("NT=512 x1536x512", 512, 1536, 512, False, 0, 4),
    ("NT=512 x2048x512", 512, 2048, 512, False, 0, 4),
    ("BQ=1536 x512x512", 1536, 512, 512, False, 0, 60),
    ("BQ=1536 x512x512 + res", 1536, 512, 512, True, 0, 20),
    ("BQ=1536 x1024x512", 1536, 1024, 512, False, 0, 6),
    ("BQ=1536 x2048x512", 1536, 2048, 512, False, 0, 6),
    ("BQ=1536 x512x2048", 1536, 512, 2048, False, 0, 6),
    ("R=9000 x512x512", 9000, 512, 512, False, 0, 4),
]


def main():
    global SHAPES
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-shapes", action="store_true")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", type=int, default=-1, help="index of the single shape to run")
    args = ap.parse_args()
    if args.train_shapes:
        SHAPES = TRAIN_SHAPES
    lib = _lib.load()
    dev = "cuda:0"
    tot_f, tot_t = 0.0, 0.0
    for si, (name, M, N, K, res, nseg, count) in enumerate(SHAPES):
        if args.only >= 0 and si != args.only:
            continue
        if nseg:
            seg = K // nseg
            src = torch.randn(M, seg, device=dev)
            idx = torch.randint(-1, M, (M, nseg), device=dev, dtype=torch.int32)
            A, lda = src, seg
        else:
            A, lda, idx, seg = torch.randn(M, K, device=dev), K, None, K
        W = torch.randn(N, K, device=dev) * 0.02
        b = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev) if res else None
        C = torch.empty(M, N, device=dev)

        def run():
            _lib.check(lib.bigc_linear_f32(_lib.current_stream(), A.data_ptr(), lda, _lib.ptr(idx), max(nseg, 1), seg, W.data_ptr(), K,
                                           b.data_ptr(), _lib.ptr(R), N, C.data_ptr(), N, M, N, K, 1))
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        fl = 2.0 * M * N * K
        print("%-22s M=%7d N=%5d K=%5d  %9.1f us  %7.1f TFLOP/s" % (name, M, N, K, ms * 1e3, fl / ms / 1e9))
        tot_f += fl * count
        tot_t += ms * count
        del A, W, C, R, idx
    print("step aggregate: %.1f GFLOP in %.2f ms = %.1f TFLOP/s" % (tot_f / 1e9, tot_t, tot_f / tot_t / 1e9))


if __name__ == "__main__":
    main()
