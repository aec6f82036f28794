#!/usr/bin/env python3
"""Host timeline of the BIG-C training step (no extra device syncs): where the host spends the step and where it waits for the GPU.
    BIGC_TRAIN_TRACE=1 python benchmarks/train_phases.py [--videos 8]"""
import argparse
import os
import sys
import time

os.environ["BIGC_TRAIN_TRACE"] = "1"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (bench re-exports make_train_job / train_step_fn from benchmarks/legs/train.py)
from bigc_hip import train_step as TS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=8)
    ap.add_argument("--n-trajs", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sync-each-step", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, props, gts, gt_duras, opt, pairs = bench.make_train_job(a, 0, dev, a.videos)
    step = bench.train_step_fn(model, props, gts, gt_duras, opt, None)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    TS.TRACE.clear()
    TS.mark(None)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        step()
        if a.sync_each_step:
            torch.cuda.synchronize()
            TS.mark("WAIT for the GPU after the step")
    torch.cuda.synchronize()
    TS.mark("WAIT for the GPU after the last step")
    total = (time.perf_counter() - t0) / a.reps
    print("step: %.2f ms" % (1e3 * total))
    for k, v in TS.TRACE.items():
        print("  %-50s %7.3f ms" % (k, 1e3 * v / a.reps))
    print("  sum of the marks %.2f ms" % (1e3 * sum(TS.TRACE.values()) / a.reps))


if __name__ == "__main__":
    main()
