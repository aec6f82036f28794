"""A Recall@50 that is not zero on synthetic data: train a BIG-C model on the HIP training step for a few hundred iterations on
the learnable relation set of ``synthetic.relational_video`` and score held-out videos of the same kind.

BASELINE's metric is "tracklet-pairs/sec + Recall@50 parity".  With SURVEY.md section 8d's random predicates and seeded random
weights every recall is 0 and the parity statement compares 0 with 0; here the ground truth is a function of the inputs (marked
actor tracklets, predicate = function of the two actors), so a few hundred steps of the reference's training loop
(tools/train_vidvrd.py:151-164: zero_grad, forward, mean, backward, clip_grad_norm_(5), Adam) bring held-out Recall@50 to 0.3 - 1.0
and the HIP eval path, its reduced-precision mode and the CPU oracle can be compared on relations that are actually found
(VidVRDhelperEvalAPIs/visual_relation_detection.py:64-120 through bigc_hip/evaluate.py)."""
import numpy as np
import torch

from .synthetic import relational_video

# a small model of the exp2 STRUCTURE for tests (two encoder / three decoder layers, I3D branch, one-layer head): trains to
# Recall@50 = 1.0 on held-out videos in ~200 steps of batch 16 (2 s on one MI355X)
CONFIG_SMALL = dict(
    num_enti_cats=9, num_pred_cats=20, dim_ffn=64, dim_enti=64, dim_pred=64, dim_att=64, dim_feat=128, dim_clsme=12, dim_i3d=32,
    enco_pool_len=4, n_enco_layers=2, n_deco_layers=3, n_att_head=4, num_querys=32, neg_weight=0.1, positive_vIoU_th=0.5,
    EntiNameEmb_path=None, bias_matrix_path=None,
    cost_coeff_dict=dict(classification=1.0, adj_matrix=30.0), loss_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
)

# the same for the VidOR variant (models/model_0v7.py: sine position embedding, classeme columns in the features, two-layer head) with
# the 81 / 51 category tables the grounding stage is hard-wired to (models/grd_model_v5.py:162,168)
CONFIG_SMALL_O = dict(
    num_enti_cats=81, num_pred_cats=51, dim_ffn=64, dim_enti=64, dim_pred=64, dim_att=64, dim_feat=128, dim_clsme=12,
    enco_pool_len=4, n_enco_layers=2, n_deco_layers=3, n_att_head=4, num_querys=32, neg_weight=0.1, positive_vIoU_th=0.5, use_clsme=True,
    EntiNameEmb_path=None, bias_matrix_path=None,
    cost_coeff_dict=dict(classification=1.0, adj_matrix=30.0), loss_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
)


def relational_set(cfg, seed0, n_videos, n_trajs=64, n_actors=4, marker=3.0, marker_width=None, feat_width=None, label_noise=0.0, motion=False):
    """``n_videos`` (proposal, gt graph) pairs, seeds ``seed0 + v``, shaped by the model config ``cfg``"""
    width = feat_width if feat_width is not None else cfg["dim_feat"] + (cfg.get("dim_i3d", None) or 0) + \
        (cfg["dim_clsme"] if cfg.get("use_clsme", False) and cfg.get("EntiNameEmb_path", None) is None and "dim_i3d" not in cfg else 0)
    mw = marker_width if marker_width is not None else max(4, min(128, cfg["dim_feat"] // (2 * n_actors)))
    return [relational_video(seed0 + v, n_trajs=n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"],
                             num_pred_cats=cfg["num_pred_cats"], n_actors=n_actors, marker=marker, marker_width=mw,
                             name="relational_%d" % (seed0 + v), label_noise=label_noise, motion=motion) for v in range(n_videos)]


def train(model, data, steps, batch=16, lr=1e-4, seed=0, log=None):
    """The reference's loop body (tools/train_vidvrd.py:151-164) on the HIP training step: ``steps`` iterations of ``batch`` videos
    drawn from ``data`` [(proposal on the model's device, gt graph on the host)].  -> last loss (float)"""
    dev = next(model.parameters()).device
    opt = torch.optim.Adam(model.parameters(), lr=lr, fused=dev.type == "cuda")
    rs = np.random.RandomState(seed)
    duras = {id(g): g.traj_durations.clone() for _, g in data}
    loss = None
    for it in range(steps):
        pick = rs.choice(len(data), size=min(batch, len(data)), replace=False)
        props, gts = [data[i][0] for i in pick], [data[i][1] for i in pick]
        for g in gts:
            g.traj_durations = duras[id(g)].clone()      # the forward decrements them in place (reference quirk, models/model_0v10.py:567)
        opt.zero_grad()
        loss, terms = model(props, gts)
        loss = loss.mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
        opt.step()
        if log is not None and it % 50 == 0:
            log("iter %d loss %.4f " % (it, float(loss)) + " ".join("%s %.4f" % (k, float(v)) for k, v in terms.items()))
    for _, g in data:
        g.traj_durations = duras[id(g)].clone()
    return float(loss.detach())


def ground_truth(held):
    """evaluator-format ground truth of held-out (proposal, graph) pairs"""
    from . import evaluate as E
    return {p.video_name: E.gt_to_eval_format(g) for p, g in held}


def relations(proposals, results):
    """evaluator-format predictions of a list of per-video model outputs"""
    from . import evaluate as E
    out = {}
    for p, r in zip(proposals, results):
        out.update(E.to_eval_format(p, r))
    return out
