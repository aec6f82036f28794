"""Recall@50 parity on a trained model (the second half of BASELINE.json's metric)."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

from .common import (EXACT_F32, F16_MFMA_PEAK_TFLOPS, F32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, ROOT, PackedBatch, TrajProposal, _lib, bigc_state_shapes,
                     build_model, collect_families, configs, device_proposal, family_rooflines, gemm_roofline, init_dist, seeded_state_dict,
                     synthetic_proposal, timed_with_events, usable_cores)


def recall_leg(args, lib, device):
    """BASELINE's metric, second half ("+ Recall@50 parity"), on relations that are actually FOUND: the exp2-dims model is trained for
    `--recall-steps` iterations of the HIP training step (the reference's loop body, tools/train_vidvrd.py:151-164) on the learnable
    synthetic relation set (bigc_hip/recall_set.py: marked actor tracklets, 12 GT relations per video as SURVEY.md 8d), then
    `--recall-videos` HELD-OUT videos are scored by the HIP eval path (default mode and the opt-in reduced-precision mode) and by the
    CPU oracle with the same trained weights, through the evaluator of VidVRDhelperEvalAPIs/visual_relation_detection.py:64-120
    (bigc_hip/evaluate.py).  Gates: |R@50(HIP) - R@50(oracle)| = 0 in parity mode, <= 1e-3 in the reduced-precision mode."""
    import copy
    from bigc_hip import evaluate as E, recall_set as R
    from oracle import bigc_oracle as O
    cfg = dict(configs.CONFIG_V)
    t0 = time.perf_counter()
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 11, att_gain=1.0)
    torch.manual_seed(1234)
    trainee = build_model(cfg, sd, device, "vidvrd", is_train=True)
    # a HARD version of the set (round 4): weaker actor markers and 30 % label noise in the ground truth (training and held-out), so that a
    # trained model's Recall@50 sits well inside (0, 1) -- about 1 - noise -- where a change of arithmetic can move it; without them every
    # mode scored 1.0 and the gate could not tell them apart (benchmarks/probes/recall_probe.py: settings tried, incl. a motion-dependent
    # predicate whose training outcome varied too much from run to run to gate on)
    hard = dict(n_actors=4, marker=1.5, label_noise=0.3)
    data = R.relational_set(cfg, 20000, 64, n_trajs=args.n_trajs, **hard)
    for p, _ in data:
        p.to(device)
    gen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    last_loss = R.train(trainee, data, args.recall_steps, batch=16, lr=1e-4)
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    sd_t = {k: v.detach().cpu() for k, v in trainee.state_dict().items()}
    del trainee, data
    torch.cuda.empty_cache()
    model = build_model(cfg, sd_t, device, "vidvrd", is_train=False)
    held = R.relational_set(cfg, 30000, args.recall_videos, n_trajs=args.n_trajs, **hard)
    gts = R.ground_truth(held)
    props = [p for p, _ in held]
    dev_props = [copy.deepcopy(p).to(device) for p in props]
    rel = {"hip": R.relations(props, model(dev_props, topk=args.topk))}
    if not EXACT_F32:
        lib.bigc_linear_set_fast(1)
        try:
            rel["hip_fast"] = R.relations(props, model(dev_props, topk=args.topk))
            lib.bigc_linear_set_fast(2)
            rel["hip_bf16"] = R.relations(props, model(dev_props, topk=args.topk))
        finally:
            lib.bigc_linear_set_fast(0)
    torch.set_num_threads(usable_cores())
    t0 = time.perf_counter()
    with torch.no_grad():
        rel["oracle"] = R.relations(props, O.forward_test(sd_t, cfg, props, args.topk))
    oracle_s = time.perf_counter() - t0
    res = {k: E.evaluate(gts, v) for k, v in rel.items()}          # (mean AP, recall {50, 100}, tagging precision)
    out = dict(trained="exp2 dims, %d iterations x 16 videos (N=%d) on the HIP training step, Adam lr 1e-4, dropout on; last loss %.3f; %.1f s (+ %.1f s of data)"
                       % (args.recall_steps, args.n_trajs, last_loss, train_s, gen_s),
               held_out_videos=len(held), gt_relations=sum(len(x) for x in gts.values()),
               recall50_oracle=res["oracle"][1][50], recall50_hip=res["hip"][1][50], recall100_oracle=res["oracle"][1][100], recall100_hip=res["hip"][1][100],
               mean_ap_oracle=res["oracle"][0], mean_ap_hip=res["hip"][0],
               relations_oracle=sum(len(x) for x in rel["oracle"].values()), relations_hip=sum(len(x) for x in rel["hip"].values()),
               abs_diff_recall50_hip_vs_oracle=abs(res["hip"][1][50] - res["oracle"][1][50]),
               relation_set="marked actors (marker 1.5), predicate = f(actor pair), 30 % of the ground-truth predicates replaced by random classes",
               recall50_inside_the_discriminating_band_0p5_0p9=bool(0.5 <= res["oracle"][1][50] <= 0.9),
               parity_gate="|R@50(HIP) - R@50(oracle)| = 0 at R@50 > 0 (SURVEY.md 8d)",
               parity_gate_passed=bool(res["oracle"][1][50] > 0 and res["hip"][1][50] == res["oracle"][1][50] and res["hip"][1][100] == res["oracle"][1][100]),
               oracle_seconds=round(oracle_s, 1))
    if "hip_fast" in res:
        out.update(recall50_hip_fast=res["hip_fast"][1][50], recall100_hip_fast=res["hip_fast"][1][100], mean_ap_hip_fast=res["hip_fast"][0],
                   abs_diff_recall50_fast_vs_parity_mode=abs(res["hip_fast"][1][50] - res["hip"][1][50]),
                   fast_gate="|R@50(reduced precision) - R@50(parity mode)| <= 1e-3 (SURVEY.md 8d)",
                   fast_gate_passed=bool(abs(res["hip_fast"][1][50] - res["hip"][1][50]) <= 1e-3))
    if "hip_bf16" in res:
        out.update(recall50_hip_bf16=res["hip_bf16"][1][50], recall100_hip_bf16=res["hip_bf16"][1][100], mean_ap_hip_bf16=res["hip_bf16"][0],
                   abs_diff_recall50_bf16_vs_parity_mode=abs(res["hip_bf16"][1][50] - res["hip"][1][50]),
                   bf16_gate_passed=bool(abs(res["hip_bf16"][1][50] - res["hip"][1][50]) <= 1e-3))
    model._engine = None
    del model
    torch.cuda.empty_cache()
    return out

