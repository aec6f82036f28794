"""The headline workload at twice the videos per step."""
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


def batch512_leg(args, lib, device, props=None):
    """The same eval step on twice the headline's videos per step (2 x --videos, default 512): from this size on the decoder's GEMM
    operands are written in the split form by their producers and the whole decoder runs on the pre-split-operand GEMM (automatic,
    csrc/model.hip; profiles/HISTORY_rounds1-4.md section 5 round 3).  Not the headline -- the default step stays at the survey's 256-video set -- but in
    the same run, so that the batch-size sensitivity is measured and not a number in a document.  `props`: the headline's device-resident
    videos, each taken twice (a throughput figure: no new host-side generation); None = 2 x --videos fresh seeds."""
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    nv = 2 * args.videos
    t0 = time.time()
    if props is None:
        props = [synthetic_proposal(70000 + v, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"]).to(device) for v in range(nv)]
        what = "fresh seeds"
    else:
        props = list(props) + list(props)
        what = "the headline's videos, each twice"
    packed = PackedBatch(props, device)
    props = None
    packed.proposals = None
    gen_s = time.time() - t0
    model = build_model(cfg, sd, device)
    pairs = packed.n_pairs()

    def step():
        return model.counts(model.forward_packed(packed, args.topk))

    steps = 5
    el, el_prof, fam = timed_with_events(lib, step, 3, steps)
    roof = gemm_roofline(fam, steps, 1e3 * el_prof / steps)
    out = dict(value=round(pairs * steps / el, 1), unit="pairs/s", videos_per_step=nv, ms_per_step=round(1e3 * el / steps, 3), steps=steps,
               roofline=roof, setup_s=round(gen_s, 1), videos=what,
               note="same workload and model as the headline at twice the videos per step; decoder GEMMs on pre-split operands (automatic from 512 videos)")
    model._engine = None
    del model, packed
    torch.cuda.empty_cache()
    return out
