#!/usr/bin/env python3
"""cProfile of the host side of the BIG-C training step (bench.py --mode train's step): python benchmarks/train_host_profile.py"""
import argparse
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (bench re-exports make_train_job / train_step_fn from benchmarks/legs/train.py)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=8)
    ap.add_argument("--n-trajs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, props, gts, gt_duras, opt, pairs = bench.make_train_job(a, 0, dev, a.videos)
    step = bench.train_step_fn(model, props, gts, gt_duras, opt, None)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(30)


if __name__ == "__main__":
    main()
