#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on one MI355X (bench.py stays the headline, configs[1]):

  --workload vidor      configs[3]: BIG-C VidOR variant (model_0v7, exp5 dims: RoI 1024 + classeme 300, 6 enc / 4 dec),
                        synthetic videos of N=180 tracklets with L in [15,1000) (SURVEY.md 8d "config 4"), topk=3
  --workload grounding  configs[2]/[3] second stage: the grounding scorer (config G) on the triplet queries the classification
                        stage produced for the same videos, clip features (T,1024), T = ceil(video_len/8); the
                        reference's rt_triplets_topk cut keeps the 200 best queries per video
Prints one JSON line per workload: pairs/s resp. triplet-queries/s, ms per video, per-kernel-family milliseconds
(HIP events inside the library)."""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from bigc_hip import _lib, configs  # noqa: E402
from bigc_hip.packed import PackedBatch  # noqa: E402
from bigc_hip.synthetic import bigc_state_shapes, grd_state_shapes, seeded_state_dict, synthetic_proposal  # noqa: E402


def families(lib, steps):
    out = {}
    _lib.check(lib.bigc_prof_collect())
    for f in range(lib.bigc_prof_family_count()):
        ms, n, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
        lib.bigc_prof_get(f, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
        if n.value:
            out[lib.bigc_prof_family_name(f).decode()] = dict(ms_per_step=round(ms.value / steps, 3), launches_per_step=n.value // steps,
                                                              tflops=round(fl.value / (ms.value * 1e-3) / 1e12, 2) if fl.value else None)
    return out


def timed(lib, fn, warmup, steps):
    for _ in range(warmup):
        fn()
    lib.bigc_prof_reset()
    lib.bigc_prof_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    lib.bigc_prof_enable(0)
    return el, r, families(lib, steps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["vidor", "grounding", "both"], default="both")
    ap.add_argument("--videos", type=int, default=16)
    ap.add_argument("--n-trajs", type=int, default=180)
    ap.add_argument("--max-len", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--max-queries", type=int, default=200)
    a = ap.parse_args()
    from models import BIG_C_vidor, DEBUG
    dev = torch.device("cuda:0")
    lib = _lib.load()
    cfg = dict(configs.CONFIG_O)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidor"), 11, att_gain=8.0)
    d = tempfile.mkdtemp()
    np.save(d + "/b.npy", sd["bias_matrix"].numpy())
    big = BIG_C_vidor(dict(cfg, bias_matrix_path=d + "/b.npy"), is_train=False)
    big.load_state_dict(sd)
    big = big.to(dev)
    width = cfg["dim_feat"] + cfg["dim_clsme"]
    t0 = time.time()
    props = [synthetic_proposal(4321 + v, n_trajs=a.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"], len_range=(15, a.max_len),
                                start_range=(0, a.max_len)).to(dev) for v in range(a.videos)]
    video_len = [p.video_len for p in props]
    packed = PackedBatch(props, dev)
    pairs = packed.n_pairs()
    rows = packed.n_rows
    props, packed.proposals = None, None
    setup = time.time() - t0

    def cls_step():
        out = big.forward_packed(packed, 3)
        out["n_out"].cpu()
        return out

    el, out, fam = timed(lib, cls_step, a.warmup, a.steps)
    if a.workload in ("vidor", "both"):
        print(json.dumps({"metric": "tracklet_pairs_per_sec", "value": round(pairs * a.steps / el, 1), "unit": "pairs/s", "n_gpus": 1,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * el / a.steps, 3), "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "VidOR classification stage (BASELINE configs[3]): BIG-C model_0v7 eval forward incl. "
                                                 "construct_triplet, exp5 dims, %d synthetic videos x N=%d, L in [15,%d), topk=3"
                                                 % (a.videos, a.n_trajs, a.max_len), "rows_per_step": rows, "pairs_per_step": pairs},
                          "ms_per_video": round(1e3 * el / a.steps / a.videos, 3), "kernel_families": fam, "setup_s": round(setup, 1)}))
    if a.workload == "vidor":
        return
    # second stage on the queries of the first
    results = big.unpack_results(a.videos, list(range(a.videos)), out)
    gcfg = dict(configs.CONFIG_G)
    gcfg.setdefault("loss_factor", {})
    gsd = seeded_state_dict(grd_state_shapes(gcfg), 12, 4.0)
    np.save(d + "/ge.npy", gsd["EntiNameEmb"].numpy())
    np.save(d + "/gp.npy", gsd["PredNameEmb"].numpy())
    grd = DEBUG(dict(gcfg, EntiNameEmb_path=d + "/ge.npy", PredNameEmb_path=d + "/gp.npy"), is_train=False)
    grd.load_state_dict(gsd)
    grd = grd.to(dev)
    jobs = []
    rs = np.random.RandomState(99)
    for v, r in enumerate(results):
        if r is None:
            continue
        quint, scores, dura = r[0], r[1], r[2]
        if quint.shape[0] > a.max_queries:                     # rt_triplets_topk (experiments/exp6/config_rt200.py:14)
            keep = torch.sort(torch.argsort(scores.mean(dim=-1), descending=True)[:a.max_queries])[0]
            quint, dura = quint[keep], dura[keep]
        T = -(-video_len[v] // 8)
        clip = torch.from_numpy((0.1 * rs.standard_normal((T, gcfg["dim_feat"]))).astype(np.float32)).to(dev)
        jobs.append((clip, quint.to(dev), dura.to(dev), video_len[v]))
    n_q = sum(int(j[1].shape[0]) for j in jobs)
    qt = sum(int(j[1].shape[0]) * int(j[0].shape[0]) for j in jobs)

    def grd_step():
        last = None
        for clip, quint, dura, vl in jobs:
            last = grd.forward_raw(clip, quint, dura, vl, 0.9, 0.5, 0.2, 0.8)
        last["status"].cpu()
        return last

    el, _, fam = timed(lib, grd_step, a.warmup, a.steps)
    print(json.dumps({"metric": "triplet_queries_per_sec", "value": round(n_q * a.steps / el, 1), "unit": "queries/s", "n_gpus": 1,
                      "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * el / a.steps, 3), "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "grounding stage (BASELINE configs[2]/[3]): grd_model_v5 DEBUG eval forward incl. temporal pooling + NMS, "
                                             "config G, %d videos, %d queries (<=%d per video), T = ceil(video_len/8) in [%d,%d]"
                                             % (len(jobs), n_q, a.max_queries, min(int(j[0].shape[0]) for j in jobs), max(int(j[0].shape[0]) for j in jobs)),
                                 "query_clip_rows_per_step": qt},
                      "ms_per_video": round(1e3 * el / a.steps / max(1, len(jobs)), 3), "kernel_families": fam}))


if __name__ == "__main__":
    main()
