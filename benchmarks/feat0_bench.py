#!/usr/bin/env python3
"""The K = 2048 front-end GEMM (fc_feat2enti.0, models/model_0v10.py:446) on rows read IN PLACE, alone: M = 579 295 rows of 2880 floats
(the 256-video benchmark batch's row count), the first 2048 columns x W (512, 2048), ReLU, split-form output -- through
bigc_gemm_ps_rows (csrc/gemm_ps.hip, A mode 3).  Prints ms / TFLOP/s per variant selected by the arguments; the tile size and the
ablations come from the environment (BIGC_PS_RB, BIGC_PS_ABLATE, BIGC_PS_BLK), which the library reads once per process.
    python benchmarks/feat0_bench.py [--two-acc 0|1] [--rows N] [--reps 10]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
from bigc_hip import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--two-acc", type=int, default=1)
ap.add_argument("--rows", type=int, default=579295)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--k", type=int, default=2048)
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--width", type=int, default=2880)
args = ap.parse_args()
lib = _lib.load()
dev = "cuda:0"
M, K, N, Wd = args.rows, args.k, args.n, args.width
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((M, Wd), generator=g, device=dev)
W = torch.randn((N, K), generator=g, device=dev) * (2.0 / (N + K)) ** 0.5
b = torch.zeros(N, device=dev)
table = (X.data_ptr() + 4 * Wd * torch.arange(M, dtype=torch.int64)).to(dev)
Cs = torch.empty((M, 4 * N), dtype=torch.uint8, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
s = _lib.current_stream()


def call():
    _lib.check(lib.bigc_gemm_ps_rows(s, table.data_ptr(), W.data_ptr(), K, b.data_ptr(), None, 0, Cs.data_ptr(), 4 * N, M, N, K, 1, args.two_acc, flag.data_ptr()))


for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    call()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.reps
print("feat0 in place: M=%d K=%d N=%d two_acc=%d env[RB=%s ABLATE=%s BLK=%s F32_DMA=%s]: %.3f ms  %.1f TFLOP/s  flag %d" % (
    M, K, N, args.two_acc, os.environ.get("BIGC_PS_RB", "-"), os.environ.get("BIGC_PS_ABLATE", "-"), os.environ.get("BIGC_PS_BLK", "-"),
    os.environ.get("BIGC_PS_F32_DMA", "-"), ms, 2.0 * M * N * K / ms / 1e9, int(flag.item())), flush=True)
