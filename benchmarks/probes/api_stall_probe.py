"""Where do the occasional slow model(proposal_list) calls come from?  Times N calls under: default, gc frozen, table upload through
the pinned staging block in <= 64 KB pieces (BIGC_UPLOAD_ASYNC_MAX raised)."""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from bigc_hip import configs
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict

dev = torch.device("cuda:0")
cfg = dict(configs.CONFIG_V)
sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
props = [bench.device_proposal(1234 + v, 64, 2880, 36, (8, 64), (0, 64), dev) for v in range(nv)]
model = bench.build_model(cfg, sd, dev)


def run(tag, n=40):
    for _ in range(3):
        model(props, topk=10)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        model(props, topk=10)
        ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.asarray(ts)
    print("%-28s median %.2f  mean %.2f  max %.2f  slow(>1.3x median): %s" % (tag, np.median(ts), ts.mean(), ts.max(), np.round(ts[ts > 1.3 * np.median(ts)], 1)), flush=True)


run("default")
gc.collect(); gc.freeze()
run("gc frozen")
gc.disable()
run("gc disabled")
gc.enable()
os.environ["BIGC_UPLOAD_ASYNC_MAX"] = str(8 << 20)
run("frozen + staged upload")
