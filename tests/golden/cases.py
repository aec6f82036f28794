"""Golden-vector case table shared by ``make_golden.py`` (which runs the real reference
in the build container) and the tests (which regenerate the same seeded weights/inputs
and compare the oracle / the HIP path against the stored reference outputs).

A case is data only: a config name, a weight seed and a synthetic-proposal spec.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_PKG = os.path.join(_ROOT, "vidsgg-big_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from bigc_hip import configs  # noqa: E402
from bigc_hip.synthetic import synthetic_proposal, bigc_state_shapes, seeded_state_dict  # noqa: E402

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))

# name -> dict(cfg, variant, wseed, topk, att_gain, videos=[proposal kwargs...])
BIGC_CASES = {
    # G1: tiny dims, every intermediate stored
    "bigc_tiny": dict(cfg="CONFIG_TINY", variant="vidvrd", wseed=11, topk=3, att_gain=6.0, full=True,
                      videos=[dict(seed=101, n_trajs=6, len_range=(5, 13), start_range=(0, 6))]),
    # tiny dims without I3D (the other concat order, models/model_0v10.py:503)
    "bigc_tiny_noi3d": dict(cfg="CONFIG_TINY_NOI3D", variant="vidvrd", wseed=12, topk=3, att_gain=6.0, full=True,
                            videos=[dict(seed=102, n_trajs=7, len_range=(5, 13), start_range=(0, 6))]),
    # G5 edge cases at tiny dims: single tracklet (=> None, no pair), Tmax=5 (conv out 3 < pool 4),
    # disjoint durations (=> None), empty video (=> None)
    "bigc_tiny_edges": dict(cfg="CONFIG_TINY", variant="vidvrd", wseed=13, topk=3, att_gain=6.0, full=False,
                            videos=[dict(seed=201, n_trajs=1, len_range=(9, 10), start_range=(0, 1)),
                                    dict(seed=202, n_trajs=5, len_range=(5, 6), start_range=(0, 3)),
                                    dict(seed=203, n_trajs=3, len_range=(5, 6), start_range=(0, 1), disjoint=True),
                                    dict(seed=204, n_trajs=0),
                                    dict(seed=205, n_trajs=9, len_range=(5, 40), start_range=(0, 30))]),
    # G2: full config V, N=64, the benchmark's first video (seed 1234)
    "bigc_v_n64": dict(cfg="CONFIG_V", variant="vidvrd", wseed=7, topk=10, att_gain=8.0, full=False,
                       videos=[dict(seed=1234, n_trajs=64)]),
    # ... and four more videos of the benchmark workload (seeds 1235..1238) through the real reference, outputs only
    "bigc_v_n64_more": dict(cfg="CONFIG_V", variant="vidvrd", wseed=7, topk=10, att_gain=8.0, full=False, slim=True,
                            videos=[dict(seed=1235 + i, n_trajs=64) for i in range(4)]),
    # ... and more VidOR-sized videos (exp5 dims) through the real reference, outputs only: N = 120 / 180 / 75 tracklets of up to 400 frames
    "bigo_o_more": dict(cfg="CONFIG_O", variant="vidor", wseed=35, topk=3, att_gain=8.0, full=False, slim=True,
                        videos=[dict(seed=2290, n_trajs=120, len_range=(15, 400), start_range=(0, 200)),
                                dict(seed=2291, n_trajs=180, len_range=(15, 120), start_range=(0, 300)),
                                dict(seed=2292, n_trajs=75, len_range=(30, 250), start_range=(0, 60))]),
    # ... and a video of LONG tracklets (400 .. 1000 frames: the upper end of the VidOR benchmark leg, bench.py, L in [15, 1000)) through the
    # real reference, outputs only: conv length 500, 41 k packed rows for one video
    "bigo_o_long": dict(cfg="CONFIG_O", variant="vidor", wseed=35, topk=3, att_gain=8.0, full=False, slim=True,
                        videos=[dict(seed=2295, n_trajs=60, len_range=(400, 1000), start_range=(0, 200))]),
    # G3: the VidOR model (models/model_0v7.py): classeme carried in the features / category embedding /
    # no classeme at all, tiny dims with all intermediates; then the exp5 dims with N=32
    "bigo_tiny_clsme": dict(cfg="CONFIG_TINY_O", variant="vidor", wseed=31, topk=3, att_gain=6.0, full=True,
                            videos=[dict(seed=501, n_trajs=8, len_range=(5, 30), start_range=(0, 10)),
                                    dict(seed=502, n_trajs=2, len_range=(5, 6), start_range=(0, 2))]),
    "bigo_tiny_emb": dict(cfg="CONFIG_TINY_O_EMB", variant="vidor", wseed=32, topk=3, att_gain=6.0, full=False,
                          videos=[dict(seed=503, n_trajs=11, len_range=(5, 20), start_range=(0, 10))]),
    "bigo_tiny_noclsme": dict(cfg="CONFIG_TINY_O_NOCLSME", variant="vidor", wseed=33, topk=3, att_gain=6.0, full=False,
                              videos=[dict(seed=504, n_trajs=7, len_range=(5, 20), start_range=(0, 10))]),
    "bigo_o_n32": dict(cfg="CONFIG_O", variant="vidor", wseed=34, topk=3, att_gain=8.0, full=False,
                       videos=[dict(seed=2234, n_trajs=32, len_range=(15, 200), start_range=(0, 100))]),
    # more than 64 tracklets per video (the decoder's un-fused role-attention branch, csrc/model.hip): exp2 dims with
    # N=100, and the VidOR model at its defining size N=180 (experiments/exp5/config_.py:40)
    "bigc_v_n100": dict(cfg="CONFIG_V", variant="vidvrd", wseed=7, topk=10, att_gain=8.0, full=False,
                        videos=[dict(seed=4100, n_trajs=100)]),
    "bigo_o_n180": dict(cfg="CONFIG_O", variant="vidor", wseed=35, topk=3, att_gain=8.0, full=False,
                        videos=[dict(seed=2280, n_trajs=180, len_range=(15, 200), start_range=(0, 150))]),
    # the other shipped experiment shapes at FULL dims through the real reference, outputs only (round 6):
    #   exp1 (experiments/exp1/config_.py): PKU RoI 2048 only, no I3D -> the head's other concat order, K = 2 x 512 + 2 x 512 + 2 x 300 + 512 (models/model_0v10.py:503)
    #   exp3 (experiments/exp3/config_.py): MEGA RoI 1024, no I3D
    #   exp4 (experiments/exp4/config_.py:21): the VidOR model without classemes (use_clsme = False)
    "bigc_exp1_n48": dict(cfg="CONFIG_V_EXP1", variant="vidvrd", wseed=41, topk=10, att_gain=8.0, full=False, slim=True,
                          videos=[dict(seed=3101, n_trajs=48), dict(seed=3102, n_trajs=20, len_range=(5, 40), start_range=(0, 20))]),
    "bigc_exp3_n40": dict(cfg="CONFIG_V_EXP3", variant="vidvrd", wseed=42, topk=10, att_gain=8.0, full=False, slim=True,
                          videos=[dict(seed=3103, n_trajs=40)]),
    "bigo_exp4_n40": dict(cfg="CONFIG_O_EXP4", variant="vidor", wseed=43, topk=3, att_gain=8.0, full=False, slim=True,
                          videos=[dict(seed=3104, n_trajs=40, len_range=(15, 200), start_range=(0, 100)),
                                  dict(seed=3105, n_trajs=90, len_range=(15, 80), start_range=(0, 150))]),
}


def get_cfg(name):
    if name == "CONFIG_TINY_NOI3D":
        c = dict(configs.CONFIG_TINY)
        c.pop("dim_i3d")
        return c
    if name == "CONFIG_TINY_O_EMB":
        c = dict(configs.CONFIG_TINY_O)
        c["_with_EntiNameEmb"] = True       # a category-embedding table instead of per-frame classeme
        return c
    if name == "CONFIG_TINY_O_NOCLSME":
        c = dict(configs.CONFIG_TINY_O)
        c["use_clsme"] = False
        return c
    if name == "CONFIG_V_EXP1":             # experiments/exp1: exp2 without the I3D columns
        c = dict(configs.CONFIG_V)
        c.pop("dim_i3d")
        return c
    if name == "CONFIG_V_EXP3":             # experiments/exp3: MEGA RoI features (1024-d), no I3D
        c = dict(configs.CONFIG_V)
        c.pop("dim_i3d")
        c["dim_feat"] = 1024
        return c
    if name == "CONFIG_O_EXP4":             # experiments/exp4: exp5 without classemes
        c = dict(configs.CONFIG_O)
        c["use_clsme"] = False
        return c
    return dict(getattr(configs, name))


def feat_width(cfg, variant):
    if variant == "vidvrd":
        return cfg["dim_feat"] + (cfg.get("dim_i3d", None) or 0)
    per_frame_clsme = cfg.get("use_clsme", False) and cfg.get("EntiNameEmb_path") is None and not cfg.get("_with_EntiNameEmb", False)
    return cfg["dim_feat"] + (cfg["dim_clsme"] if per_frame_clsme else 0)


def make_proposal(cfg, variant, spec):
    spec = dict(spec)
    disjoint = spec.pop("disjoint", False)
    p = synthetic_proposal(feat_width=feat_width(cfg, variant), num_enti_cats=cfg["num_enti_cats"], **spec)
    if disjoint and p.num_proposals > 0:
        # place the tracklets one after the other so that no two overlap in time
        t = 0
        for i in range(p.num_proposals):
            n = p.bboxes_list[i].shape[0]
            p.traj_durations[i, 0] = t
            p.traj_durations[i, 1] = t + n - 1
            t += n + 2
        p.video_len = t
    return p


# ---- grounding scorer (models/grd_model_v5.py) ------------------------------------------------
# thresholds of experiments/grounding_weights/config_.py:74-80
GRD_TH = dict(score_th=0.9, tiou_th=0.5, bins_th=0.2, nms_th=0.8)
GRD_CASES = {
    # G4 tiny (hidden 64 = 8 heads x 8, 3 bins, T=9, n=5): every intermediate stored
    "grd_tiny": dict(cfg=dict(dim_feat=32, dim_clsme=300, dim_hidden=64, num_bins=3), wseed=41, gain=4.0, full=True,
                     videos=[dict(seed=601, T=9, n=5, video_len=70), dict(seed=602, T=33, n=1, video_len=260),
                             dict(seed=603, T=2, n=4, video_len=16)]),
    # full dims (config G), n=64 queries x T=120 clips: outputs only
    "grd_g_n64": dict(cfg=dict(dim_feat=1024, dim_clsme=300, dim_hidden=128, num_bins=10), wseed=42, gain=4.0, full=False,
                      videos=[dict(seed=604, T=120, n=64, video_len=960)]),
    # config G on more videos (VidOR-like clip counts T = ceil(video_len / 8), up to 200 queries), final outputs only
    "grd_g_more": dict(cfg=dict(dim_feat=1024, dim_clsme=300, dim_hidden=128, num_bins=10), wseed=42, gain=4.0, full=False, slim=True,
                       videos=[dict(seed=611, T=75, n=200, video_len=600), dict(seed=612, T=240, n=37, video_len=1920),
                               dict(seed=613, T=33, n=120, video_len=260), dict(seed=614, T=300, n=12, video_len=2400)]),
}


def make_grd_inputs(cfg, spec):
    """Synthetic grounding input of one video: clip features 0.1*N(0,1) (T,dim_feat),
    n triplet queries [pred, s_cat, o_cat, s_tid, o_tid] with an overlap span inside the video."""
    import numpy as np
    import torch
    rs = np.random.RandomState(spec["seed"])
    T, n, video_len = spec["T"], spec["n"], spec["video_len"]
    feat = torch.from_numpy((0.1 * rs.standard_normal((T, cfg["dim_feat"]))).astype(np.float32))
    quint = np.stack([rs.randint(1, 51, n), rs.randint(1, 81, n), rs.randint(1, 81, n), rs.randint(0, 50, n), rs.randint(0, 50, n)], 1)
    s = rs.randint(0, video_len // 2, n)
    e = s + rs.randint(1, video_len // 2, n)
    dura = np.stack([s, e], 1)
    return feat, (torch.from_numpy(quint.astype(np.int64)), torch.from_numpy(dura.astype(np.int64)), video_len)


def make_grd_case(name):
    """-> (cfg, state_dict, [(video_feature, (quintuples, dura_inters, video_len)), ...])"""
    from bigc_hip.synthetic import grd_state_shapes
    case = GRD_CASES[name]
    cfg = dict(case["cfg"])
    cfg["loss_factor"] = dict(classification=1.0, centerness=1.0, regression=1.0)
    sd = seeded_state_dict(grd_state_shapes(cfg), case["wseed"], case["gain"])
    return cfg, sd, [make_grd_inputs(cfg, v) for v in case["videos"]]


# ---- BIG-C training step (models/model_0v10.py:510-704) -------------------------------------------
TRAIN_CASES = {
    # G6 tiny: a batch of two videos, gradients of EVERY parameter stored
    "bigc_train_tiny": dict(cfg="CONFIG_TINY", variant="vidvrd", wseed=51, att_gain=3.0, full=True,
                            videos=[dict(seed=701, n_trajs=9, len_range=(6, 25), start_range=(0, 8)),
                                    dict(seed=702, n_trajs=6, len_range=(5, 12), start_range=(0, 4))],
                            gts=[dict(seed=711, n_gt_trajs=5, n_preds=4), dict(seed=712, n_gt_trajs=4, n_preds=3)]),
    "bigo_train_tiny": dict(cfg="CONFIG_TINY_O", variant="vidor", wseed=52, att_gain=3.0, full=True,
                            videos=[dict(seed=703, n_trajs=8, len_range=(6, 25), start_range=(0, 8))],
                            gts=[dict(seed=713, n_gt_trajs=5, n_preds=4)]),
    # exp2 dims, one N=16 video: loss terms, matching, gradient norms of every parameter + a few full gradients
    "bigc_train_v": dict(cfg="CONFIG_V", variant="vidvrd", wseed=53, att_gain=3.0, full=False,
                         videos=[dict(seed=704, n_trajs=16)], gts=[dict(seed=714, n_gt_trajs=6, n_preds=5)]),
    # exp5 dims (model_0v7: RoI 1024 + classeme 300, 6 encoder / 4 decoder layers), two videos of N = 14 / 9: VidOR training step
    "bigo_train_o": dict(cfg="CONFIG_O", variant="vidor", wseed=54, att_gain=3.0, full=False,
                         videos=[dict(seed=705, n_trajs=14), dict(seed=706, n_trajs=9)],
                         gts=[dict(seed=715, n_gt_trajs=6, n_preds=5), dict(seed=716, n_gt_trajs=4, n_preds=3)]),
}
# the benchmark-sized training step through the real reference: two videos of N = 64 / 40 at exp2 dims, loss terms, matching and the
# gradient NORM of every parameter (no full gradients: a few KB)
TRAIN_CASES["bigc_train_v64"] = dict(cfg="CONFIG_V", variant="vidvrd", wseed=53, att_gain=3.0, full=False, slim=True,
                                     videos=[dict(seed=707, n_trajs=64), dict(seed=708, n_trajs=40)],
                                     gts=[dict(seed=717, n_gt_trajs=8, n_preds=10), dict(seed=718, n_gt_trajs=6, n_preds=6)])
TRAIN_FULL_GRADS_V = ["fc_pred2logits.bias", "decoder_layers.5.norm3.weight", "decoder_layers.0.fc_pred2att.bias",
                      "encoder_layers.0.norm1.weight", "conv_feat2enti.bias", "fc_bbox2enti.0.weight", "pos_embedding"]


def make_train_case(name):
    """-> (cfg, variant, state_dict, [proposals], [gt_graphs])"""
    from bigc_hip.synthetic import synthetic_gt_graph
    case = TRAIN_CASES[name]
    cfg = get_cfg(case["cfg"])
    sd = seeded_state_dict(bigc_state_shapes(cfg, case["variant"]), case["wseed"], case.get("att_gain", 1.0))
    props = [make_proposal(cfg, case["variant"], v) for v in case["videos"]]
    gts = [synthetic_gt_graph(p, num_pred_cats=cfg["num_pred_cats"], **g) for p, g in zip(props, case["gts"])]
    return cfg, case["variant"], sd, props, gts


def make_case(name):
    """-> (cfg, variant, state_dict, [proposals], topk)"""
    case = BIGC_CASES[name]
    cfg = get_cfg(case["cfg"])
    sd = seeded_state_dict(bigc_state_shapes(cfg, case["variant"]), case["wseed"], case.get("att_gain", 1.0))
    props = [make_proposal(cfg, case["variant"], v) for v in case["videos"]]
    return cfg, case["variant"], sd, props, case["topk"]


# ---- evaluator (VidVRDhelperEvalAPIs) ----------------------------------------------------------------
EVAL_SEEDS = [901, 902, 903]


def make_eval_case(seed, n_videos=6):
    """Seeded groundtruth / prediction relation sets in the evaluator's dict format: some predictions are jittered
    copies of GT relations (hits with vIoU above / below 0.5), some have a wrong triplet, some duplicate a GT."""
    import numpy as np
    rs = np.random.RandomState(seed)
    ents = ["e%d" % i for i in range(5)]
    preds = ["p%d" % i for i in range(4)]

    def traj(n, x0, y0):
        xy = np.stack([x0 + np.arange(n) * rs.uniform(-2, 2), y0 + np.arange(n) * rs.uniform(-2, 2)], 1)
        wh = rs.uniform(20, 80, size=(1, 2))
        return np.concatenate([xy, xy + wh], 1)

    gt, pr = {}, {}
    for v in range(n_videos):
        vid = "video_%d" % v
        g_list, p_list = [], []
        for _ in range(rs.randint(0, 7) if v else 5):
            s = int(rs.randint(0, 30))
            n = int(rs.randint(5, 40))
            rel = {"triplet": [ents[rs.randint(5)], preds[rs.randint(4)], ents[rs.randint(5)]], "duration": (s, s + n),
                   "sub_traj": traj(n, rs.uniform(0, 300), rs.uniform(0, 200)).tolist(),
                   "obj_traj": traj(n, rs.uniform(0, 300), rs.uniform(0, 200)).tolist()}
            g_list.append(rel)
            for _ in range(rs.randint(0, 4)):
                jit = rs.choice([2.0, 10.0, 40.0])
                shift = int(rs.randint(-3, 4))
                s2 = max(0, s + shift)
                n2 = max(1, n - abs(shift))
                st = (np.asarray(rel["sub_traj"])[:n2] + rs.uniform(-jit, jit, size=(1, 4))).tolist()
                ot = (np.asarray(rel["obj_traj"])[:n2] + rs.uniform(-jit, jit, size=(1, 4))).tolist()
                trip = list(rel["triplet"])
                if rs.uniform() < 0.2:
                    trip[1] = preds[rs.randint(4)]
                p_list.append({"triplet": trip, "duration": (s2, s2 + n2), "score": float(rs.uniform()), "sub_traj": st, "obj_traj": ot})
        for _ in range(rs.randint(0, 6)):
            n = int(rs.randint(5, 20))
            p_list.append({"triplet": [ents[rs.randint(5)], preds[rs.randint(4)], ents[rs.randint(5)]], "duration": (0, n),
                           "score": float(rs.uniform()), "sub_traj": traj(n, 10, 10).tolist(), "obj_traj": traj(n, 200, 100).tolist()})
        gt[vid] = g_list
        if v != 2:                     # one video has no prediction entry at all
            pr[vid] = p_list
    return gt, pr


# ---- grounding training step (models/grd_model_v5.py:375-527) -----------------------------------------
GRD_TRAIN_CASES = {
    "grd_train_tiny": dict(cfg=dict(dim_feat=32, dim_clsme=300, dim_hidden=64, num_bins=3), wseed=81, gain=2.0, rng_seed=4242, full=True,
                           videos=[dict(seed=801, T=11, n_gt_trajs=5, n_preds=5), dict(seed=802, T=6, n_gt_trajs=4, n_preds=3)]),
    "grd_train_g": dict(cfg=dict(dim_feat=1024, dim_clsme=300, dim_hidden=128, num_bins=10), wseed=82, gain=2.0, rng_seed=4343, full=False,
                        videos=[dict(seed=803, T=40, n_gt_trajs=6, n_preds=8)]),
}
GRD_TRAIN_FULL_GRADS = ["regr_head.4.point_wise.weight", "combined_encoder.norme.weight", "proj2sim.weight", "temp_fc.weight",
                        "video_encoder.convs.0.depth_wise.weight", "query_fc.bias", "PredNameEmb"]


def make_grd_train_case(name):
    """-> (cfg, state_dict, [clip features], [gt graphs], rng_seed). GT predicates get random sub-spans of the subject/object
    overlap as their durations and the first predicate is duplicated with another span (a multi-instance query)."""
    import numpy as np
    import torch
    from bigc_hip.synthetic import grd_state_shapes, synthetic_gt_graph
    case = GRD_TRAIN_CASES[name]
    cfg = dict(case["cfg"])
    cfg["loss_factor"] = dict(classification=1.0, centerness=1.0, regression=1.0)
    sd = seeded_state_dict(grd_state_shapes(cfg), case["wseed"], case["gain"])
    feats, gts = [], []
    for v in case["videos"]:
        rs = np.random.RandomState(v["seed"])
        p = synthetic_proposal(v["seed"], n_trajs=10, feat_width=4, num_enti_cats=81, len_range=(30, 64), start_range=(0, 20))
        g = synthetic_gt_graph(p, v["seed"] + 1, n_gt_trajs=v["n_gt_trajs"], n_preds=v["n_preds"], num_pred_cats=51)
        pd = g.pred_durations.clone()
        for k in range(pd.shape[0]):
            s, e = int(pd[k, 0]), int(pd[k, 1])
            a = rs.randint(s, e + 1)
            b = rs.randint(a, e + 1)
            pd[k, 0], pd[k, 1] = a, b
        dup = pd[:1].clone()
        dup[0, 0] = g.pred_durations[0, 0]
        g.pred_durations = torch.cat([pd, dup], 0)
        g.pred_cat_ids = torch.cat([g.pred_cat_ids, g.pred_cat_ids[:1]], 0)
        g.adj_matrix = torch.cat([g.adj_matrix, g.adj_matrix[:, :1, :]], 1)
        g.num_preds = int(g.pred_cat_ids.shape[0])
        gts.append(g)
        feats.append(torch.from_numpy((0.1 * rs.standard_normal((v["T"], cfg["dim_feat"]))).astype(np.float32)))
    return cfg, sd, feats, gts, case["rng_seed"]


# --------------------------------------------------------------------------------------------------------------------
# f2: tracking-result ingest (dataloaders/dataloader_vidvrd.py:479-700).  Synthetic `.npy` contents in the PKU row format
# [fid, tid, x,y,w,h (tracker), conf, cat, x,y,w,h (detector), feat...] (tracking_results/format_demo.py:9-21).
# --------------------------------------------------------------------------------------------------------------------
INGEST_CASES = {
    "ingest_pku_i3d": dict(seed=901, n_traj=9, dim_box=16, dim_i3d=8, min_frames_th=5, max_proposal=6, with_i3d=True),
    "ingest_pku": dict(seed=902, n_traj=7, dim_box=12, dim_i3d=0, min_frames_th=4, max_proposal=50, with_i3d=False),
    # more draws of the same generator: more tracklets than max_proposal (sort by score + truncation), a threshold that filters several,
    # wider features, no truncation at all
    "ingest_pku_i3d_b": dict(seed=903, n_traj=14, dim_box=24, dim_i3d=12, min_frames_th=7, max_proposal=5, with_i3d=True),
    "ingest_pku_i3d_c": dict(seed=904, n_traj=6, dim_box=8, dim_i3d=4, min_frames_th=3, max_proposal=100, with_i3d=True),
    "ingest_pku_b": dict(seed=905, n_traj=12, dim_box=20, dim_i3d=0, min_frames_th=10, max_proposal=4, with_i3d=False),
    "ingest_pku_c": dict(seed=906, n_traj=16, dim_box=16, dim_i3d=0, min_frames_th=5, max_proposal=9, with_i3d=False),
}


def make_ingest_case(name):
    """-> (track_res (n_box, 12+dim_box) float32, i3d rows (n_box, 12+dim_i3d) float32 or None, case dict).
    Tracklets with gaps (to be interpolated), undetected boxes (cat <= 0: tracker box only, zero features), mixed
    categories (majority vote), one short tracklet (filtered) and one all-background tracklet."""
    return make_pku_rows(INGEST_CASES[name])


def make_pku_rows(case):
    """the generator of make_ingest_case for any case dict (tests/test_tools_gpu.py writes full-width PKU files with it)"""
    import numpy as np
    rs = np.random.RandomState(case["seed"])
    rows = []
    for t in range(case["n_traj"]):
        tid = 3 * t + 1                                           # ids need not be consecutive
        start = rs.randint(0, 20)
        n = rs.randint(6, 18) if t != 2 else 3                    # tracklet 2 is shorter than min_frames_th
        fids = start + np.sort(rs.choice(np.arange(n + 6), size=n, replace=False))   # gaps of up to a few frames
        main_cat = rs.randint(1, 30)
        for j, fid in enumerate(fids):
            xy, wh = rs.uniform(0, 500, 2), rs.uniform(20, 120, 2)
            conf = rs.uniform(0.3, 1.0)
            cat = main_cat if rs.rand() < 0.8 else rs.randint(1, 30)
            if t == 4:
                cat = 0                                            # never detected: background tracklet
            elif rs.rand() < 0.15:
                cat = -1                                           # this box has no detection
            dxy, dwh = xy + rs.uniform(-5, 5, 2), wh + rs.uniform(-5, 5, 2)
            feat = np.maximum(rs.standard_normal(case["dim_box"]), 0)      # ReLU-like: exact zeros occur
            rows.append(np.concatenate([[fid, tid], xy, wh, [conf, cat], dxy, dwh, feat]))
    order = np.argsort([r[0] for r in rows], kind="stable")       # file order: by frame, tracklets interleaved
    track = np.stack([rows[i] for i in order]).astype(np.float32)
    i3d = None
    if case["with_i3d"]:
        i3d = np.concatenate([track[:, :12], np.maximum(rs.standard_normal((track.shape[0], case["dim_i3d"])), 0)], 1).astype(np.float32)
    return track, i3d, case


# VidOR tracking results (dataloaders/dataloader_vidor_v3.py:389-485): rows of length 12 + dim_box or of length 6 (tracker-only boxes,
# skipped by the loader), a classeme row per file row, score clipping before the top-N cut.
VIDOR_INGEST_CASES = {
    # ragged file (6-column rows inside tracklets and one tracklet made of them only), gaps, score threshold that removes some
    "ingest_vidor": dict(seed=911, n_traj=10, dim_box=16, dim_clsme=12, min_frames_th=5, max_proposal=6, score_th=0.55, ragged=True),
    # dense 2-D file, more tracklets than max_proposal, a threshold below every score
    "ingest_vidor_b": dict(seed=912, n_traj=14, dim_box=24, dim_clsme=300, min_frames_th=7, max_proposal=5, score_th=0.2, ragged=False),
    # high threshold: few survive (fewer than max_proposal), long gaps
    "ingest_vidor_c": dict(seed=913, n_traj=8, dim_box=8, dim_clsme=20, min_frames_th=3, max_proposal=40, score_th=0.68, ragged=True),
}


def make_vidor_ingest_case(name):
    """-> (track_res: list of rows (Python lists; fid / tid / category are ints as in the reference's pickled object arrays),
            classeme (n_file_rows, dim_clsme) float32, case dict)."""
    import numpy as np
    case = VIDOR_INGEST_CASES[name]
    rs = np.random.RandomState(case["seed"])
    rows = []
    for t in range(case["n_traj"]):
        tid = 5 * t + 2
        start = rs.randint(0, 25)
        n = rs.randint(6, 20) if t != 2 else 3                    # tracklet 2 is shorter than min_frames_th
        fids = start + np.sort(rs.choice(np.arange(n + 8), size=n, replace=False))
        main_cat = rs.randint(1, 80)
        base_conf = rs.uniform(0.35, 0.95)                         # per-tracklet confidence level: the score threshold separates tracklets
        for fid in fids:
            xy, wh = rs.uniform(0, 500, 2), rs.uniform(20, 120, 2)
            conf = float(np.clip(base_conf + rs.uniform(-0.1, 0.1), 0.01, 0.99))
            cat = main_cat if rs.rand() < 0.8 else rs.randint(1, 80)
            if t == 4:
                cat = 0                                            # majority category 0: background tracklet
            short = case["ragged"] and (t == 6 or rs.rand() < 0.12)     # tracklet 6: tracker-only boxes only; elsewhere: a gap
            if short:
                rows.append([int(fid), int(tid)] + [float(np.float32(v)) for v in np.concatenate([xy, wh])])
                continue
            dxy, dwh = xy + rs.uniform(-5, 5, 2), wh + rs.uniform(-5, 5, 2)
            feat = np.maximum(rs.standard_normal(case["dim_box"]), 0)
            vals = [float(np.float32(v)) for v in np.concatenate([xy, wh, [conf]])]
            rows.append([int(fid), int(tid)] + vals[:4] + [vals[4], int(cat)] + [float(np.float32(v)) for v in np.concatenate([dxy, dwh, feat])])
    order = np.argsort([r[0] for r in rows], kind="stable")       # file order: by frame, tracklets interleaved
    track = [rows[i] for i in order]
    classeme = rs.standard_normal((len(track), case["dim_clsme"])).astype(np.float32)
    return track, classeme, case


# --------------------------------------------------------------------------------------------------------------------
# f3: the explicit pairwise baseline Base_C (models/model_pairwise_baseline.py)
# --------------------------------------------------------------------------------------------------------------------
BASEC_CASES = {
    # tiny dims, classeme columns in the features (EntiNameEmb_path None), all triplets returned
    "basec_tiny": dict(cfg=dict(num_enti_cats=9, num_pred_cats=7, dim_feat=24, dim_clsme=12, dim_enti=16, dim_ffn=32, enco_pool_len=4,
                                use_clsme=True, rt_triplets_topk=-1, EntiNameEmb_path=None), wseed=91, topk=3,
                       videos=[dict(seed=9101, n_trajs=7, len_range=(5, 14), start_range=(0, 8)), dict(seed=9102, n_trajs=1),
                               dict(seed=9103, n_trajs=0), dict(seed=9104, n_trajs=5, len_range=(4, 9), start_range=(0, 30))]),
    # category embedding as classeme, top-20 cut
    "basec_tiny_emb": dict(cfg=dict(num_enti_cats=9, num_pred_cats=7, dim_feat=24, dim_clsme=12, dim_enti=16, dim_ffn=32, enco_pool_len=4,
                                    use_clsme=True, rt_triplets_topk=20, _with_EntiNameEmb=True), wseed=92, topk=2,
                           videos=[dict(seed=9201, n_trajs=8, len_range=(5, 14), start_range=(0, 8))]),
    # exp6 dimensions (VidOR: 81/51 categories, 1024-d RoI + 300-d classeme), N=24
    "basec_o_n24": dict(cfg=dict(num_enti_cats=81, num_pred_cats=51, dim_feat=1024, dim_clsme=300, dim_enti=512, dim_ffn=512, enco_pool_len=4,
                                 use_clsme=True, rt_triplets_topk=200, EntiNameEmb_path=None), wseed=93, topk=3,
                        videos=[dict(seed=9301, n_trajs=24, len_range=(15, 70), start_range=(0, 40))]),
}


# Base_C training step (models/model_pairwise_baseline.py:276-312): labelled pairs + multi-hot predicate labels per video
BASEC_TRAIN_CASES = {
    "basec_train_tiny": dict(base="basec_tiny", videos=[0, 3], pairs=[9, 6], lseed=9701),
    "basec_train_tiny_emb": dict(base="basec_tiny_emb", videos=[0], pairs=[14], lseed=9702),
    "basec_train_o_n24": dict(base="basec_o_n24", videos=[0], pairs=[60], lseed=9703),
}


def make_basec_train_case(name):
    """-> (cfg, state_dict, [proposals], [pairid2trajids (n,2) int64], [multihot (n,P) float32])"""
    import numpy as np
    import torch
    case = BASEC_TRAIN_CASES[name]
    cfg, sd, props, _ = make_basec_case(case["base"])
    rs = np.random.RandomState(case["lseed"])
    props = [props[i] for i in case["videos"]]
    pos, labels = [], []
    for p, n_pairs in zip(props, case["pairs"]):
        n = p.num_proposals
        allp = np.array([(s, o) for s in range(n) for o in range(n) if s != o], dtype=np.int64)
        sel = allp[rs.permutation(len(allp))[:n_pairs]]                       # distinct ordered pairs, random order
        lab = np.zeros((len(sel), cfg["num_pred_cats"]), dtype=np.float32)
        for r in range(len(sel)):                                             # 1-3 positive predicates per pair (never background)
            lab[r, rs.choice(np.arange(1, cfg["num_pred_cats"]), size=rs.randint(1, 4), replace=False)] = 1.0
        pos.append(torch.from_numpy(sel))
        labels.append(torch.from_numpy(lab))
    return cfg, sd, props, pos, labels


def make_basec_case(name):
    """-> (cfg, state_dict, [proposals], topk)"""
    from bigc_hip.synthetic import basec_state_shapes
    case = BASEC_CASES[name]
    cfg = dict(case["cfg"])
    sd = seeded_state_dict(basec_state_shapes(cfg), case["wseed"], 1.0)
    width = cfg["dim_feat"] + (cfg["dim_clsme"] if (cfg["use_clsme"] and not cfg.get("_with_EntiNameEmb")) else 0)
    props = []
    for v in case["videos"]:
        kw = dict(n_trajs=v["n_trajs"], feat_width=width, num_enti_cats=cfg["num_enti_cats"])
        if "len_range" in v:
            kw.update(len_range=v["len_range"], start_range=v["start_range"])
        props.append(synthetic_proposal(v["seed"], **kw))
    return cfg, sd, props, case["topk"]


# --------------------------------------------------------------------------------------------------------------------
# f1: result conversion (utils/evaluate.py:72-151).  Seeded model outputs in the 4-tuple format of BIG_C.forward.
# --------------------------------------------------------------------------------------------------------------------
def make_evalfmt_cases():
    """-> {name: (dataset_type, use_pku, proposal, result)} with result = None or (quintuples i64 (n,5), scores f32 (n,3),
    dura_inters i64 (n,2)): subject / object pairs with overlapping durations, spans that are sub-intervals of the overlap (as after
    grounding), background predicates (skipped by the converter), both VidVRD category numberings, a VidOR video name."""
    import numpy as np
    import torch
    out = {}
    for name, dataset, use_pku, seed, n_trajs, n_rel, n_ecat, n_pcat, vname in (
            ("vidvrd", "vidvrd", False, 9901, 9, 40, 36, 133, "ILSVRC2015_train_00005015"),
            ("vidvrd_pku", "vidvrd", True, 9902, 7, 25, 36, 133, "ILSVRC2015_train_00010001"),
            ("vidor", "vidor", False, 9903, 12, 30, 81, 51, "0001_3598080384")):
        rs = np.random.RandomState(seed)
        p = synthetic_proposal(seed, n_trajs=n_trajs, feat_width=4, num_enti_cats=n_ecat, len_range=(10, 40), start_range=(0, 12), name=vname)
        du = p.traj_durations.numpy()
        rows, spans = [], []
        while len(rows) < n_rel:
            s, o = rs.randint(0, n_trajs, 2)
            lo, hi = max(du[s, 0], du[o, 0]), min(du[s, 1], du[o, 1])
            if s == o or lo > hi:
                continue
            a = rs.randint(lo, hi + 1)
            b = rs.randint(a, hi + 1)
            pred = 0 if rs.rand() < 0.15 else rs.randint(1, n_pcat)
            rows.append([pred, int(p.cat_ids[s]), int(p.cat_ids[o]), s, o])
            spans.append([a, b] if rs.rand() < 0.5 else [lo, hi])
        scores = torch.from_numpy(rs.uniform(0, 1, size=(n_rel, 3)).astype(np.float32))
        out[name] = (dataset, use_pku, p, (torch.tensor(rows, dtype=torch.long), scores, torch.tensor(spans, dtype=torch.long)))
    out["vidvrd_none"] = ("vidvrd", False, synthetic_proposal(9904, n_trajs=3, feat_width=4, num_enti_cats=36, name="ILSVRC2015_train_00000001"), None)
    return out


# --------------------------------------------------------------------------------------------------------------------
# G5: exact probability ties inside construct_triplet (models/model_0v10.py:712-713).  torch.topk's choice among EXACTLY equal
# probabilities is an artefact of its partial sort; the fixture records what the reference does so that the build's documented
# rule (lowest class id) is pinned against it.
# --------------------------------------------------------------------------------------------------------------------
def make_ties_case():
    """-> (proposal, logits (nq, P) f32, att (2, nq, N) f32, topk, boundary_ties {query: tied class ids})
    queries 0-3: ties INSIDE the top-k (the same k classes whichever order); queries 4-7: a tie that straddles position k (several
    classes compete for the last slots); the remaining queries have distinct values."""
    import numpy as np
    import torch
    rs = np.random.RandomState(77)
    nq, P, topk, N = 12, 9, 3, 6
    p = synthetic_proposal(411, n_trajs=N, feat_width=8, num_enti_cats=7, len_range=(20, 30), start_range=(0, 5))
    logits = np.stack([rs.permutation(P) for _ in range(nq)]).astype(np.float32) * 0.5 - 2.0
    logits[0, [2, 5, 7]] = 4.0                     # three-way tie filling the top 3
    logits[1, [1, 8]] = 3.5; logits[1, 4] = 5.0    # tie for ranks 2-3
    logits[2, [0, 3]] = 6.0; logits[2, 6] = 2.5    # tie for ranks 1-2 (class 0 = background among them)
    logits[3, [6, 7, 8]] = 3.0
    boundary = {}
    logits[4, :] = -3.0; logits[4, 1] = 5.0; logits[4, 2] = 4.0; logits[4, [3, 5, 8]] = 3.0      # 3 classes tie for the third slot
    boundary[4] = [3, 5, 8]
    logits[5, :] = -1.0; logits[5, 7] = 2.0; logits[5, [0, 4, 6, 2]] = 1.0                       # 4 classes tie for slots 2 and 3
    boundary[5] = [0, 2, 4, 6]
    logits[6, :] = 0.25                                                                          # everything ties
    boundary[6] = list(range(P))
    logits[7, :] = -2.0; logits[7, [1, 2]] = 3.0; logits[7, [5, 6]] = 1.5                        # 2 classes tie for the last slot
    boundary[7] = [5, 6]
    att = torch.zeros(2, nq, N)
    so = rs.randint(0, N, size=(2, nq))
    so[1] = (so[0] + 1 + rs.randint(0, N - 1, size=nq)) % N            # subject != object
    for r in range(2):
        att[r, np.arange(nq), so[r]] = 1.0
    return p, torch.from_numpy(logits), att, topk, boundary
