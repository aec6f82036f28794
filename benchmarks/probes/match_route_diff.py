#!/usr/bin/env python3
"""per-parameter gradient difference between the host-matched and the device-matched training step (tests/test_hip_matching.py)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests/golden", "tests", "vidsgg-big_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import cases  # noqa: E402
from test_hip_train import build_train_model  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bigc_train_tiny"
cfg, variant, sd, props, gts = cases.make_train_case(name)
model = build_train_model(cfg, variant, sd)
duras = [g.traj_durations.clone() for g in gts]


def step(host_match, host_up):
    os.environ["BIGC_MATCH_ON_HOST"] = host_match
    os.environ["BIGC_UPSTREAM_ON_HOST"] = host_up
    for g, d in zip(gts, duras):
        g.traj_durations = d.clone()
    model.zero_grad()
    total, terms = model(props, gts)
    total.backward()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


for a, b in ((("1", "1"), ("1", "1")), (("1", "1"), ("1", "0")), (("1", "1"), ("0", "1")), (("0", "0"), ("0", "0"))):
    g0, g1 = step(*a), step(*b)
    bad = {k: float((g0[k] - g1[k]).abs().max() / g0[k].abs().max().clamp_min(1e-30)) for k in g0 if not torch.equal(g0[k], g1[k])}
    print(a, "vs", b, ":", len(bad), "of", len(g0), "differ;", sorted(bad.items(), key=lambda kv: -kv[1])[:4])
