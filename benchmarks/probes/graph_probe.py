#!/usr/bin/env python3
"""Does a hipGraph of the eval step (148 launches) beat stream launches?  python benchmarks/probes/graph_probe.py [videos]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
import bench  # noqa: E402
from bigc_hip import configs  # noqa: E402
from bigc_hip.packed import PackedBatch  # noqa: E402
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_proposal  # noqa: E402


def main():
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    props = [synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"]).to(dev) for v in range(nv)]
    packed = PackedBatch(props, dev)
    model = bench.build_model(cfg, sd, dev)

    def step():
        return model.forward_packed(packed, 10)

    for _ in range(3):
        out = step()
    torch.cuda.synchronize()

    def timeit(fn, n=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    print("stream launches: %.3f ms per step" % timeit(step))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gout = step()
    torch.cuda.synchronize()
    print("graph replay:    %.3f ms per step" % timeit(g.replay))
    ref = step()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(ref[k], gout[k]) for k in ("pred_logits", "n_out") if k in ref)
    print("graph output identical to stream output:", same, sorted(ref.keys()))


if __name__ == "__main__":
    main()
