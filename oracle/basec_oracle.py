"""CPU oracle for the explicit pairwise baseline (TEST INFRASTRUCTURE ONLY).

Restates ``Base_C.forward_propagation`` / ``prediction_head`` / ``construct_triplet`` / ``_forward_train`` + ``loss`` of
``models/model_pairwise_baseline.py`` (:174-197, :236-273, :306-395, :276-312) with stock torch CPU ops, the front-end being the
one of ``oracle.bigc_oracle`` (same layers, :176-189).  Pinned by ``tests/golden/basec_*.npz``."""
import numpy as np
import torch
import torch.nn.functional as F

from . import bigc_oracle as O


def pair_ids(n):
    """trajid2pairid (:105-112): all ordered pairs without the diagonal, row-major"""
    mask = torch.ones(n, n, dtype=torch.bool)
    mask.fill_diagonal_(False)
    return mask.nonzero(as_tuple=False)


def forward_propagation(sd, cfg, proposal, pairs=None):
    """(:174-197, 236-273) -> (pair logits (n_pairs, P), enti2enco (N, de)); pairs = pairid2trajids, default all ordered pairs"""
    dim_feat, n = cfg["dim_feat"], proposal.num_proposals
    traj_bboxes, traj_features = O.preprocess_proposal(proposal)
    visual, classeme = traj_features[:, :, :dim_feat], traj_features[:, :, dim_feat:]
    fb = O.linear(O.linear(traj_bboxes, sd, "fc_bbox2enti.0", True), sd, "fc_bbox2enti.2", True)
    fv = O.linear(O.linear(visual, sd, "fc_feat2enti.0", True), sd, "fc_feat2enti.2", True)
    enti = torch.cat([fb, fv], -1).permute(0, 2, 1)
    conv = F.conv1d(enti, sd["conv_feat2enti.weight"], sd["conv_feat2enti.bias"], stride=2, padding=1)
    pooled = F.adaptive_max_pool1d(conv, cfg["enco_pool_len"]).reshape(n, -1)
    feat = O.linear(O.linear(pooled, sd, "fc_enti2enco.0", True), sd, "fc_enti2enco.2", True)
    pairs = pair_ids(n) if pairs is None else pairs
    cats = proposal.cat_ids[pairs]
    bias = sd["bias_matrix"][cats[:, 0], cats[:, 1], :]
    parts = [feat[pairs[:, 0]], feat[pairs[:, 1]]]
    if cfg["use_clsme"]:
        clsme = sd["EntiNameEmb"][proposal.cat_ids] if "EntiNameEmb" in sd else classeme.mean(dim=1)
        parts = [clsme[pairs[:, 0]], clsme[pairs[:, 1]]] + parts
    h = O.linear(torch.cat(parts, -1), sd, "fc_pred2logits.0", True)
    return O.linear(h, sd, "fc_pred2logits.2") + bias, feat


def construct_triplet(proposal, pred_logits, topk, rt_triplets_topk):
    """(:306-395)"""
    n = proposal.num_proposals
    probs = torch.softmax(pred_logits, dim=-1)
    scores, cats = torch.topk(probs, topk, dim=-1)
    scores, cats = scores.reshape(-1), cats.reshape(-1)
    so = torch.repeat_interleave(pair_ids(n), topk, dim=0)
    inter, ok = O.duration_overlap(proposal.traj_durations, proposal.traj_durations)
    ok = ok.clone()
    ok.fill_diagonal_(False)
    pos = ok[so[:, 0], so[:, 1]]
    if int(pos.sum()) == 0:
        return None
    so, scores, cats = so[pos], scores[pos], cats[pos]
    quint = torch.cat([cats[:, None], proposal.cat_ids[so], so], -1)
    trip = torch.cat([scores[:, None], proposal.scores[so]], -1)
    q = quint.numpy()
    order = torch.from_numpy(np.lexsort(tuple(q[:, c] for c in range(4, -1, -1))))       # unique rows: torch.unique(dim=0) = a sort
    quint, trip = quint[order], trip[order]
    dura = inter[quint[:, 3], quint[:, 4], :]
    mask = quint[:, 0] != 0
    quint, trip, dura = quint[mask], trip[mask], dura[mask]
    if rt_triplets_topk > 0:
        top = trip.mean(dim=-1).argsort(descending=True)[:rt_triplets_topk]
        quint, trip, dura = quint[top], trip[top], dura[top]
    return quint, trip, dura


def forward_test(sd, cfg, proposal_list, topk=10):
    out = []
    for p in proposal_list:
        if p.num_proposals == 0:
            out.append(None)
            continue
        logits, _ = forward_propagation(sd, cfg, p)
        out.append(construct_triplet(p, logits, topk, cfg["rt_triplets_topk"]))
    return out


def forward_train(sd, cfg, proposal_list, pos_id_list, label_list):
    """``Base_C._forward_train`` + ``loss`` (:276-312): BCE-with-logits mean over the labelled pairs of the whole batch.
    ``sd`` may hold tensors with requires_grad for autograd.  -> (total, {"cls": loss}, logits)"""
    logits = torch.cat([forward_propagation(sd, cfg, p, ids)[0] for p, ids in zip(proposal_list, pos_id_list)], 0)
    labels = torch.cat(list(label_list), 0)
    cls = F.binary_cross_entropy_with_logits(logits, labels, reduction="mean")
    return cls, {"cls": cls}, logits
