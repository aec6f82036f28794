"""CPU oracle for the grounding scorer's training step (TEST INFRASTRUCTURE ONLY).

Restates ``DEBUG._forward_train`` of ``models/grd_model_v5.py`` (:375-389) with ``prepare_gt_data``
(:253-306, including the random negative predicates drawn with ``torch.randperm`` from the GLOBAL torch RNG,
:285-303), ``get_gt_labels`` (:224-250), ``map2bins`` (:391-442) and ``loss`` (:444-527).  The network is
``oracle.grd_oracle.forward_propagation`` (differentiable), so autograd gives the reference gradients.
Parity needs the same global seed before the call (``torch.manual_seed``) because of the unseeded negatives.
Pinned by ``tests/golden/grd_train_*.npz``.
"""
import torch
import torch.nn.functional as F

from . import grd_oracle as G


def unique_rows(x):
    """``unique_with_idx_nd`` (utils/utils_func.py:330-345): lexicographically sorted unique rows and, per unique
    row, the ascending indices of its occurrences."""
    uniq, inverse = torch.unique(x, dim=0, return_inverse=True)
    return uniq, [torch.nonzero(inverse == i, as_tuple=True)[0] for i in range(uniq.shape[0])]


def prepare_gt_queries(sd, gt_graph, train=True, n_pred_cats=51):
    """(:253-306) -> words_emb (2U,3,300), temporal_info (2U,2), target (n_pred,2) normalised, index_map,
    plus the raw query tags (quintuple-like rows [pred, s_cat, o_cat] and spans) for the HIP path's tests."""
    video_len = gt_graph.video_len
    so = torch.argmax(gt_graph.adj_matrix, dim=-1).t()
    so_cats = gt_graph.traj_cat_ids[so]
    sub_d, obj_d = gt_graph.traj_durations[so[:, 0]], gt_graph.traj_durations[so[:, 1]]
    inter = torch.stack([torch.maximum(sub_d[:, 0], obj_d[:, 0]), torch.minimum(sub_d[:, 1], obj_d[:, 1])], -1)
    tags = torch.cat([gt_graph.pred_cat_ids[:, None], so_cats, inter], -1)           # [pred, s_cat, o_cat, s, e]
    uniq, index_map = unique_rows(tags)
    pred_ids = uniq[:, 0]
    if train:
        _, groups = unique_rows(uniq[:, 1:])
        all_cats = torch.arange(n_pred_cats)
        neg = []
        for im in groups:                                    # one randperm per (s_cat, o_cat, span) group, in unique order
            used = torch.zeros(n_pred_cats, dtype=torch.bool).scatter_(0, tags[im, 0], True)
            other = all_cats[~used]
            neg.append(other[torch.randperm(len(other))][:len(im)])
        neg = torch.cat(neg)
        pred_ids = torch.cat([pred_ids, neg])
        cats = torch.cat([uniq[:, 1:3], uniq[:, 1:3]], 0)
        spans = torch.cat([uniq[:, 3:], uniq[:, 3:]], 0)
    else:
        cats, spans = uniq[:, 1:3], uniq[:, 3:]
    words = torch.stack([sd["EntiNameEmb"][cats[:, 0]], sd["PredNameEmb"][pred_ids], sd["EntiNameEmb"][cats[:, 1]]], 1)
    info = spans.float() / video_len
    target = gt_graph.pred_durations.float() / video_len
    return words, info, target, index_map, torch.cat([pred_ids[:, None], cats], -1), spans


def gt_labels(target, n_clips, num_bins):
    """(:224-250)"""
    clip = torch.linspace(0, 1, n_clips)
    bins = torch.linspace(0, 1, num_bins + 1)
    ct = target.mean(dim=-1)
    bin_ids = (ct[:, None] - bins[None, :] > 0).sum(dim=-1) - 1
    left = clip[None, :] - target[:, 0, None]
    right = target[:, 1, None] - clip[None, :]
    outside = (left <= 0) | (right <= 0)
    ctness = torch.sqrt(torch.minimum(left, right) / torch.maximum(left, right))
    ctness[outside] = 0
    scores = torch.ones_like(ctness)
    scores[outside] = 0
    return torch.stack([left, right], -1), ctness, scores, bin_ids


def map_to_bins(regrs, conf, cls, bin_ids, index_map, num_bins):
    """(:391-442)"""
    n2, T, _ = regrs.shape
    n_uniq = n2 // 2
    r = regrs.reshape(n2, T, 2, num_bins)
    pc, nc, pl, nl, pr = [], [], [], [], []
    for i, imp in enumerate(index_map):
        b = bin_ids[imp]
        used = torch.zeros(num_bins, dtype=torch.bool).scatter_(0, b, True)
        pc.append(conf[i][:, b]); nc.append(conf[i][:, ~used])
        pl.append(cls[i][:, b]); nl.append(cls[i][:, ~used])
        pr.append(r[i][:, :, b])
    pos_conf, neg_conf = torch.cat(pc, -1).t(), torch.cat(nc, -1).t()
    pos_cls, neg_cls = torch.cat(pl, -1).t(), torch.cat(nl, -1).t()
    pos_regr = torch.cat(pr, -1).permute(2, 0, 1)
    neg_conf = torch.cat([neg_conf, conf[n_uniq:].permute(0, 2, 1).reshape(n_uniq * num_bins, T)], 0)
    neg_cls = torch.cat([neg_cls, cls[n_uniq:].permute(0, 2, 1).reshape(n_uniq * num_bins, T)], 0)
    return pos_conf, neg_conf, pos_cls, neg_cls, pos_regr


def loss_terms(mapped, labels, index_maps, loss_factor):
    """(:444-527)"""
    pcs, ncs, pls, nls, prs, grs, gcs, gss = [], [], [], [], [], [], [], []
    for (pc, nc, pl, nl, pr), (gr, gc, gs, _), im in zip(mapped, labels, index_maps):
        order = torch.cat(im)
        pcs.append(pc.reshape(-1)); ncs.append(nc.reshape(-1)); pls.append(pl.reshape(-1)); nls.append(nl.reshape(-1))
        prs.append(pr.reshape(-1, 2))
        grs.append(gr[order].reshape(-1, 2)); gcs.append(gc[order].reshape(-1)); gss.append(gs[order].reshape(-1))
    pc, nc, pl, nl = torch.cat(pcs), torch.cat(ncs), torch.cat(pls), torch.cat(nls)
    pr, gr, gc, gs = torch.cat(prs), torch.cat(grs), torch.cat(gcs), torch.cat(gss)
    mask = gc > 0
    pos_cls = F.binary_cross_entropy_with_logits(pl, gs)
    neg_cls = F.binary_cross_entropy_with_logits(nl, torch.zeros_like(nl))
    neg_ct = F.binary_cross_entropy_with_logits(nc, torch.zeros_like(nc))
    if int(mask.sum()) == 0:
        pos_ct, regr = torch.zeros_like(pos_cls), torch.zeros_like(pos_cls)
    else:
        pos_ct = F.binary_cross_entropy_with_logits(pc[mask], gc[mask])
        a, b = pr[mask], gr[mask]
        tiou = (torch.minimum(a[:, 1], b[:, 1]) + torch.minimum(a[:, 0], b[:, 0])) / (torch.maximum(a[:, 1], b[:, 1]) + torch.maximum(a[:, 0], b[:, 0]))
        regr = (-(tiou + 1e-6).log()).mean()
    f = loss_factor
    return {"pos_cls": pos_cls * f["classification"], "neg_cls": neg_cls * f["classification"], "pos_ct": pos_ct * f["centerness"],
            "neg_ct": neg_ct * f["centerness"], "regr": regr * f["regression"]}


def forward_train(sd, cfg, video_feature_list, gt_graph_list, drops=None):
    """(:375-389) -> (total, terms, per-video query descriptions)"""
    k = cfg["num_bins"]
    mapped, labels, maps, queries = [], [], [], []
    for feat, g in zip(video_feature_list, gt_graph_list):
        words, info, target, index_map, tags, spans = prepare_gt_queries(sd, g, True)
        regrs, conf, cls = G.forward_propagation(sd, feat, words, info, drop=None if drops is None else drops[len(mapped)])
        lab = gt_labels(target, regrs.shape[1], k)
        mapped.append(map_to_bins(regrs, conf, cls, lab[-1], index_map, k))
        labels.append(lab)
        maps.append(index_map)
        queries.append((tags, spans, target, index_map, lab[-1]))
    terms = loss_terms(mapped, labels, maps, cfg["loss_factor"])
    return torch.stack(list(terms.values())).sum(), terms, queries
