"""CPU oracle for the temporal grounding scorer (TEST INFRASTRUCTURE ONLY).

Torch-CPU fp32 restatement of ``models/grd_model_v5.py`` (class ``DEBUG``) of
Dawn-LX/VidSGG-BIG, eval path: ``prepare_data`` (:310-328), ``forward_propagation``
(:331-373) with ``QANetEncoderLayer`` (:81-137), ``DepthWiseSeparableConv1d`` (:36-56) and
``PosEncoder`` (:58-78), ``_forward_test_single`` (:530-576), ``temporal_pooling``
(:697-737) and ``temporal_nms`` / ``_nms`` (:667-695), as pure functions over a
``state_dict``.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.

Pinning: the reference has no tests for this path; the oracle is pinned against outputs of
the reference itself (``tests/golden/make_golden.py`` -> ``tests/golden/grd_*.npz``,
torch 2.10.0) in ``tests/test_oracle_golden.py``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .bigc_oracle import layer_norm, mha


def pos_table(length, d):
    """(d, length) sine table of ``PosEncoder`` (:58-78), evaluated in float32 like the reference."""
    i = np.arange(d)
    even = (i % 2) == 0
    freq = np.where(even, 10000.0 ** (-i / d), -(10000.0 ** ((1 - i) / d)))
    phase = np.where(even, 0.0, np.pi / 2)
    f = torch.from_numpy(freq.astype(np.float32))[:, None]
    ph = torch.from_numpy(phase.astype(np.float32))[:, None]
    pos = torch.arange(length, dtype=torch.float32)[None, :].repeat(d, 1)
    return torch.sin(pos * f + ph)


def dwsep(x, sd, p):
    """depth-wise (groups=C, padding=k//2) then point-wise conv, x (N,C,T)  (:36-56)"""
    w = sd[p + ".depth_wise.weight"]
    y = F.conv1d(x, w, sd[p + ".depth_wise.bias"], padding=w.shape[-1] // 2, groups=x.shape[1])
    return F.conv1d(y, sd[p + ".point_wise.weight"], sd[p + ".point_wise.bias"])


def _ln_ct(x, sd, p):
    return layer_norm(x.transpose(1, 2), sd, p).transpose(1, 2)


def qanet_layer(x, sd, p, num_conv=4, nhead=8, drop=None, base=0, p_base=0.1):
    """QANetEncoderLayer.forward, x (N,d,T)  (:110-137).  Without ``drop`` the dropouts are inactive (eval mode); with an
    ``oracle.dropout_ref.Dropper`` they are applied as the HIP path does: sites base + {0,1: after conv blocks 1 and 3 with
    p = p_base (i+1)/num_conv (:123-125), 2: attention probabilities, 3: after the attention residual (:130), 4: output (:136)},
    element index = (n T + t) d + c."""
    d, T = x.shape[1], x.shape[2]

    def dropped(t_ndT, site, prob):
        if drop is None:
            return t_ndT
        saved = drop.p
        drop.p = prob
        y = drop.rows(site, t_ndT.transpose(1, 2).reshape(-1, d), 0).reshape(-1, T, d).transpose(1, 2)
        drop.p = saved
        return y

    out = x + pos_table(T, d)[None]
    res = out
    out = _ln_ct(out, sd, p + ".normb")
    for i in range(num_conv):
        out = torch.relu(dwsep(out, sd, p + ".convs.%d" % i)) + res
        if (i + 1) % 2 == 0:
            out = dropped(out, base + i // 2, p_base * (i + 1) / num_conv)
        res = out
        out = _ln_ct(out, sd, p + ".norm_seq.%d" % i)
    seq = out.permute(2, 0, 1)                          # (T,N,d): attention over time, batch N
    if drop is not None:
        saved, drop.p = drop.p, p_base
    att = torch.stack([mha(seq[:, n], seq[:, n], seq[:, n], sd, p + ".mh_attn", nhead, drop, base + 2, n * T) for n in range(seq.shape[1])], 1)
    if drop is not None:
        drop.p = saved
    out = dropped(att.permute(1, 2, 0) + res, base + 3, p_base)
    res = out
    out = layer_norm(out.transpose(1, 2), sd, p + ".norme")
    out = torch.relu(F.linear(out, sd[p + ".fc.weight"], sd[p + ".fc.bias"])).transpose(1, 2)
    return dropped(out + res, base + 4, p_base)


def prepare_data(sd, quintuples, dura_inters, video_len):
    """-> words_emb (n,3,300) ordered [subject, predicate, object], temporal_info (n,2)  (:310-328)"""
    pred = sd["PredNameEmb"][quintuples[:, 0]]
    sub = sd["EntiNameEmb"][quintuples[:, 1]]
    obj = sd["EntiNameEmb"][quintuples[:, 2]]
    return torch.stack([sub, pred, obj], 1), dura_inters.float() / video_len


def head(x, sd, p, sigmoid=False):
    for i in range(4):
        x = torch.relu(dwsep(x, sd, p + ".%d.0" % i))
    x = dwsep(x, sd, p + ".4")
    return torch.sigmoid(x) if sigmoid else x


def forward_propagation(sd, video_feature, words_emb, inter_dura, keep=False, drop=None):
    """(:331-373) -> regrs (n,T,2k), conf_logits (n,T,k), cls_logits (n,T,k)"""
    n = words_emb.shape[0]
    video = F.linear(video_feature, sd["video_fc.weight"], sd["video_fc.bias"]).t()[None]       # (1,d,T)
    words = F.linear(words_emb, sd["query_fc.weight"], sd["query_fc.bias"]).permute(0, 2, 1)     # (n,d,3)
    temporal = F.linear(inter_dura, sd["temp_fc.weight"], sd["temp_fc.bias"])                    # (n,d)
    query = words + temporal[:, :, None]
    video = qanet_layer(video, sd, "video_encoder", drop=drop, base=0x100)
    query = qanet_layer(query, sd, "query_encoder", drop=drop, base=0x200)
    vt = video.transpose(1, 2)                                                                    # (1,T,d)
    sim = torch.matmul(F.linear(vt, sd["proj2sim.weight"]).expand(n, -1, -1), query)               # (n,T,3)
    sim_r = torch.softmax(sim, dim=2)
    sim_c = torch.softmax(sim, dim=1)
    sim_rc = torch.matmul(sim_r, sim_c.transpose(1, 2))                                           # (n,T,T)
    v = vt.expand(n, -1, -1)
    mat_a = torch.matmul(sim_r, query.transpose(1, 2))
    mat_b = torch.matmul(sim_rc, v)
    comb = torch.cat([v, mat_a, mat_a * v, mat_b * v], -1)
    comb = F.linear(comb, sd["vq_fc.weight"], sd["vq_fc.bias"]).transpose(1, 2)                   # (n,d,T)
    enc = qanet_layer(comb, sd, "combined_encoder", drop=drop, base=0x300)
    regrs = head(enc, sd, "regr_head", sigmoid=True).transpose(1, 2)
    conf = head(enc, sd, "conf_head").transpose(1, 2)
    cls = head(enc, sd, "cls_head").transpose(1, 2)
    if keep:
        return regrs, conf, cls, dict(video_enc=video[0].t(), query_enc=query.transpose(1, 2), sim=sim,
                                      combined_in=comb.transpose(1, 2), combined_enc=enc.transpose(1, 2))
    return regrs, conf, cls


def tiou_table(spans):
    """``tIoU(d, d)`` of utils/utils_func.py:375-390 (zero where the two spans do not overlap)."""
    s, e = spans[:, 0], spans[:, 1]
    ok = (e[:, None] >= s[None, :]) & (e[None, :] >= s[:, None])
    t = (torch.minimum(e[:, None], e[None, :]) - torch.maximum(s[:, None], s[None, :])) / \
        (torch.maximum(e[:, None], e[None, :]) - torch.minimum(s[:, None], s[None, :]))
    t[~ok] = 0
    return t


def nms_keep(spans, probs, th):
    """Greedy 1-D NMS in descending probability; returns the bool keep mask  (:667-695)"""
    order = probs.argsort().tolist()
    t = tiou_table(spans)
    keep = torch.zeros(len(order), dtype=torch.bool)
    while order:
        i = order.pop()
        keep[i] = True
        order = [j for j in order if bool(t[i, j] < th)]
    return keep


def temporal_pooling(regrs, scores, num_bins, score_th, tiou_th):
    """(:697-737) vectorised over (query, bin): spans of the clips whose score exceeds
    ``score_th * top`` and whose gIoU with the top clip's span exceeds ``tiou_th``, merged."""
    n, T, _ = scores.shape
    r = regrs.reshape(n, T, 2, num_bins)
    clip = torch.linspace(0, 1, T)
    start = clip[None, :, None] - r[:, :, 0, :]
    end = clip[None, :, None] + r[:, :, 1, :]
    top_score, top_id = torch.max(scores, dim=1)                   # (n,k)
    m1 = scores > score_th * top_score[:, None, :]
    ts = start.gather(1, top_id[:, None, :])
    te = end.gather(1, top_id[:, None, :])
    giou = (torch.minimum(te, end) - torch.maximum(ts, start)) / (torch.maximum(te, end) - torch.minimum(ts, start))
    m = m1 & (giou > tiou_th)
    assert bool(m.any(dim=1).all()), "empty pooling set (the reference raises here too, :726)"
    ps = torch.where(m, start, torch.full_like(start, float("inf"))).min(dim=1)[0]
    pe = torch.where(m, end, torch.full_like(end, float("-inf"))).max(dim=1)[0]
    return torch.stack([ps, pe], -1)                               # (n,k,2)


def forward_test_single(sd, video_feature, words_emb, inter_dura, num_bins, score_th, tiou_th, bins_th, nms_th):
    """(:530-576) -> pooled_se (n,k+1,2) f32, bins_probs (n,k+1) f32, bins_mask (n,k+1) bool"""
    regrs, conf, cls = forward_propagation(sd, video_feature, words_emb, inter_dura)
    scores = conf.sigmoid() * cls.sigmoid()
    probs = torch.cat([scores.max(dim=1)[0], torch.ones(scores.shape[0], 1)], 1)
    mask = probs > bins_th
    pooled = temporal_pooling(regrs, scores, num_bins, score_th, tiou_th)
    s = torch.maximum(inter_dura[:, None, 0], pooled[:, :, 0])
    e = torch.minimum(inter_dura[:, None, 1], pooled[:, :, 1])
    ov = s <= e                                                    # (n,k)
    pooled = torch.where(ov[:, :, None], torch.stack([s, e], -1), inter_dura[:, None, :].expand(-1, num_bins, -1))
    pooled = torch.cat([pooled, inter_dura[:, None, :]], 1)
    ov = torch.cat([ov, torch.ones(ov.shape[0], 1, dtype=torch.bool)], 1)
    nms = torch.stack([nms_keep(pooled[i], probs[i], nms_th) for i in range(pooled.shape[0])], 0)
    mask = mask & ov & nms
    empty = ~mask.any(dim=1)
    if bool(empty.any()):
        rows = empty.nonzero(as_tuple=True)[0]
        mask[rows, probs[rows].max(dim=-1)[1]] = True
    weak = probs[:, :-1].max(-1)[0] <= bins_th
    probs = probs.clone()
    probs[weak, -1] = 0.0
    return pooled, probs, mask


def forward(sd, cfg, video_feature, data, score_th=0.5, tiou_th=0.5, bins_th=0.1, nms_th=0.5):
    """``DEBUG.forward(..., with_gt_data=False)`` for one video (:196-221)."""
    quintuples, dura_inters, video_len = data
    if quintuples is None or quintuples.shape[0] == 0:
        return None, None
    words, info = prepare_data(sd, quintuples, dura_inters, video_len)
    return forward_test_single(sd, video_feature, words, info, cfg["num_bins"], score_th, tiou_th, bins_th, nms_th)
