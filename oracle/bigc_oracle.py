"""CPU oracle for the BIG-C classification path (TEST INFRASTRUCTURE ONLY).

This file is a CPU restatement (stock PyTorch fp32 CPU ops + numpy) of the reference's
BIG-C eval forward -- ``models/model_0v10.py`` (VidVRD) and ``models/model_0v7.py``
(VidOR) of Dawn-LX/VidSGG-BIG -- written as pure functions over a ``state_dict``.  It
follows the reference's op sequence *as written* (tracklets stretched to ``Tmax`` and
stacked, per-video loop), so it doubles as the ``cpu_baseline`` (kind "port") of
``bench.py``.

It is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The HIP path under
``vidsgg-big_amd/`` never does.

Pinning: the reference holds no tests or golden vectors for this path (SURVEY.md
section 4), so this oracle is pinned against outputs of the reference itself, run in the
build container by ``tests/golden/make_golden.py`` with numpy-seeded weights and inputs
and committed as ``tests/golden/*.npz`` (checked by ``tests/test_oracle_golden.py``).
Third-party arithmetic the reference delegates to (torch ``nn.MultiheadAttention``,
``nn.Conv1d``, ``nn.LayerNorm``, ``F.adaptive_max_pool1d``, ``torch.topk``,
``torch.unique``) is restated explicitly below; the goldens were produced with
torch 2.10.0 (recorded in each fixture).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# a1/a2: per-tracklet box features and the temporal stretch
# --------------------------------------------------------------------------------------

def stretch_index(L, M):
    """Source frame read by each of the M output frames when a length-L tracklet is
    stretched to M frames (reference: ``stack_with_repeat_2d`` models/model_0v10.py:18-46).

    The reference repeats frame c ``q+1`` times for c < r and ``q`` times otherwise,
    where ``q, r = divmod(M, L)`` (it lays M ones then zeros row-major into L columns and
    sums the columns, :37-42).
    """
    q, r = divmod(M, L)
    reps = np.full(L, q, dtype=np.int64)
    reps[:r] += 1
    return np.repeat(np.arange(L, dtype=np.int64), reps)


def bbox8(boxes, w, h):
    """(L,4) xyxy pixel boxes -> (L,8) [cx,dcx,cy,dcy,w,dw,h,dh]; the d* columns are
    forward differences with a zero in the LAST frame
    (reference: models/model_0v10.py:401-420 + ``stack_with_padding`` utils/utils_func.py:49-91)."""
    b = boxes.clone()
    b[:, 0:4:2] /= w
    b[:, 1:4:2] /= h
    cx = (b[:, 2] + b[:, 0]) / 2
    cy = (b[:, 3] + b[:, 1]) / 2
    bw = b[:, 2] - b[:, 0]
    bh = b[:, 3] - b[:, 1]
    cols = []
    for v in (cx, cy, bw, bh):
        d = torch.zeros_like(v)
        d[:-1] = v[1:] - v[:-1]
        cols += [v, d]
    return torch.stack(cols, dim=1)


def preprocess_proposal(proposal):
    """-> traj_bboxes (N,Tmax,8), traj_features (N,Tmax,D)  (models/model_0v10.py:391-430)."""
    w, h = proposal.video_wh
    lens = [b.shape[0] for b in proposal.bboxes_list]
    tmax = max(lens)
    tb, tf = [], []
    for boxes, feats in zip(proposal.bboxes_list, proposal.features_list):
        idx = torch.from_numpy(stretch_index(boxes.shape[0], tmax))
        tb.append(bbox8(boxes, w, h)[idx])
        tf.append(feats[idx])
    return torch.stack(tb, 0), torch.stack(tf, 0)


# --------------------------------------------------------------------------------------
# restated torch.nn building blocks
# --------------------------------------------------------------------------------------

def linear(x, sd, prefix, relu=False):
    y = F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))
    return torch.relu(y) if relu else y


def layer_norm(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def mha(xq, xk, xv, sd, prefix, nhead, drop=None, site=0, row0=0):
    """``nn.MultiheadAttention`` forward for one unbatched sequence, eval mode.

    xq (Lq,d), xk/xv (Lk,d); packed in_proj [q;k;v], scale 1/sqrt(head_dim), softmax
    over keys, concat heads, out_proj (reference call sites: models/model_0v10.py:109,183;
    models/grd_model_v5.py:105)."""
    d = xq.shape[-1]
    hd = d // nhead
    w, b = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = F.linear(xq, w[:d], b[:d]).reshape(-1, nhead, hd).transpose(0, 1)
    k = F.linear(xk, w[d:2 * d], b[d:2 * d]).reshape(-1, nhead, hd).transpose(0, 1)
    v = F.linear(xv, w[2 * d:], b[2 * d:]).reshape(-1, nhead, hd).transpose(0, 1)
    att = torch.softmax(torch.matmul(q, k.transpose(1, 2)) / math.sqrt(hd), dim=-1)
    if drop is not None:                                   # nn.MultiheadAttention(dropout=0.1) in training mode
        att = drop.att(site, att, row0)
    o = torch.matmul(att, v).transpose(0, 1).reshape(-1, d)
    return F.linear(o, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


def encoder_layer(x, sd, p, nhead, drop=None, layer=0, row0=0):
    """Post-norm transformer encoder layer over the N tracklets of one video
    (models/model_0v10.py:103-117).  ``drop`` (oracle.dropout_ref.Dropper) applies the training-mode dropouts of :109-114
    at sites 16*layer + {0 attention, 1 dropout1, 2 ffn, 3 dropout2}; ``row0`` = global index of the video's first tracklet."""
    a = mha(x, x, x, sd, p + ".self_attn", nhead, drop, 16 * layer, row0)
    if drop is not None:
        a = drop.rows(16 * layer + 1, a, row0)
    x = layer_norm(x + a, sd, p + ".norm1")
    h = linear(x, sd, p + ".linear1", relu=True)
    if drop is not None:
        h = drop.rows(16 * layer + 2, h, row0)
    y = linear(h, sd, p + ".linear2")
    if drop is not None:
        y = drop.rows(16 * layer + 3, y, row0)
    return layer_norm(x + y, sd, p + ".norm2")


def decoder_layer(query, pos, enco, sd, p, nhead, dim_enti, dim_att, drop=None, layer=0, row0=0, att_row0=0):
    """Role-aware decoder layer (models/model_0v10.py:178-225). Returns (query, att (2,Q,N)).  ``drop``: training-mode dropouts at
    sites 4096 + 16*layer + {0 self-attention probabilities (query rows from ``att_row0``), 1 fc2's Dropout (rows from ``row0``)}."""
    qk = query + pos
    query = layer_norm(query + mha(qk, qk, query, sd, p + ".self_attn", nhead, drop, 4096 + 16 * layer, att_row0), sd, p + ".norm1")
    query = query + pos                                   # persistent (:189)
    e2a = linear(enco, sd, p + ".fc_enti2att")
    p2a = linear(query, sd, p + ".fc_pred2att")
    half = dim_att // 2
    scores = torch.stack([
        torch.matmul(p2a[:, :half], e2a[:, :half].t()),
        torch.matmul(p2a[:, half:], e2a[:, half:].t()),
    ], 0) / np.sqrt(dim_enti)                             # sqrt(dim_enti), not sqrt(half) (:201)
    att = torch.softmax(scores, dim=2) * torch.softmax(scores, dim=0)
    role = 0
    for r in range(2):
        vals = torch.matmul(att[r], enco)
        fr = p + ".fc_rolewise.%d" % r
        role = role + linear(linear(vals, sd, fr + ".0", relu=True), sd, fr + ".2")
    query = layer_norm(query + role, sd, p + ".norm2")
    h = linear(query, sd, p + ".fc2.0", relu=True)
    if drop is not None:
        h = drop.rows(4096 + 16 * layer + 1, h, row0)
    y = linear(h, sd, p + ".fc2.3")
    query = layer_norm(query + y, sd, p + ".norm3")
    return query, att


# --------------------------------------------------------------------------------------
# a3..a7: encode2decode
# --------------------------------------------------------------------------------------

def encode2decode(sd, cfg, proposal, variant="vidvrd", keep=False, drop=None, trk0=0, vid=0):
    """models/model_0v10.py:434-475 / models/model_0v7.py:434-481. Returns dict with at
    least ``pred_logits`` (Q,P) and ``att`` (2,Q,N); with ``keep`` every intermediate."""
    out = {}
    dim_feat, de = cfg["dim_feat"], cfg["dim_enti"]
    nhead = cfg["n_att_head"]
    n = proposal.num_proposals
    traj_bboxes, traj_features = preprocess_proposal(proposal)
    visual = traj_features[:, :, :dim_feat]
    extra = traj_features[:, :, dim_feat:]                 # I3D (vidvrd) or classeme (vidor)
    fb = linear(linear(traj_bboxes, sd, "fc_bbox2enti.0", True), sd, "fc_bbox2enti.2", True)
    fv = linear(linear(visual, sd, "fc_feat2enti.0", True), sd, "fc_feat2enti.2", True)
    enti = torch.cat([fb, fv], -1).permute(0, 2, 1)        # (N, 2de, Tmax)
    conv = F.conv1d(enti, sd["conv_feat2enti.weight"], sd["conv_feat2enti.bias"], stride=2, padding=1)
    pooled = F.adaptive_max_pool1d(conv, cfg["enco_pool_len"]).reshape(n, -1)   # channel-major (:457)
    enti2enco = linear(linear(pooled, sd, "fc_enti2enco.0", True), sd, "fc_enti2enco.2", True)
    x = enti2enco
    enc_states = []
    for i in range(cfg["n_enco_layers"]):
        x = encoder_layer(x, sd, "encoder_layers.%d" % i, nhead, drop, i, trk0)
        enc_states.append(x)
    enco = x
    query, pos = sd["pred_query_init"], sd["pos_embedding"]
    dec_states, att = [], None
    for i in range(cfg["n_deco_layers"]):
        # the HIP path evaluates the first layer's self-attention block once for the whole batch (rows 0..Q-1): its attention
        # masks are shared by the videos; later layers index the batch's query rows vid*Q + q
        nq = query.shape[0]
        query, att = decoder_layer(query, pos, enco, sd, "decoder_layers.%d" % i, nhead, de, cfg["dim_att"], drop, i, vid * nq,
                                   0 if i == 0 else vid * nq)
        dec_states.append((query, att))

    # prediction head (models/model_0v10.py:478-507, models/model_0v7.py:483-513)
    cat_ids = proposal.cat_ids
    soid = torch.argmax(att, dim=-1)                        # (2,Q)
    socat = cat_ids[soid]
    bias = sd["bias_matrix"][socat[0], socat[1], :]
    sub_feat, obj_feat = enti2enco[soid[0]], enti2enco[soid[1]]   # pre-encoder features (:492)
    if variant == "vidvrd":
        sub_c, obj_c = sd["EntiNameEmb"][socat[0]], sd["EntiNameEmb"][socat[1]]
        if cfg.get("dim_i3d", None):
            i3d_avg = extra.mean(dim=1)
            sub_i = linear(i3d_avg[soid[0]], sd, "fc_i3d.0", True)
            obj_i = linear(i3d_avg[soid[1]], sd, "fc_i3d.0", True)
            head_in = torch.cat([query, sub_i, obj_i, sub_feat, obj_feat, sub_c, obj_c], -1)
        else:
            head_in = torch.cat([query, sub_c, obj_c, sub_feat, obj_feat], -1)
        logits = linear(head_in, sd, "fc_pred2logits") + bias
    else:
        if cfg["use_clsme"]:
            if "EntiNameEmb" in sd:
                sub_c, obj_c = sd["EntiNameEmb"][socat[0]], sd["EntiNameEmb"][socat[1]]
            else:
                clsme_avg = extra.mean(dim=1)
                sub_c, obj_c = clsme_avg[soid[0]], clsme_avg[soid[1]]
            head_in = torch.cat([query, sub_c, obj_c, sub_feat, obj_feat], -1)
        else:
            head_in = torch.cat([query, sub_feat, obj_feat], -1)
        logits = linear(linear(head_in, sd, "fc_pred2logits.0", True), sd, "fc_pred2logits.2") + bias
    out["pred_logits"], out["att"], out["pred_query"] = logits, att, query
    if keep:
        out.update(traj_bboxes=traj_bboxes, enti_features=enti.permute(0, 2, 1), conv_out=conv.permute(0, 2, 1),
                   pooled=pooled, enti2enco=enti2enco, enco_output=enco, enc_states=enc_states,
                   dec_states=dec_states, pred_soid=soid, head_in=head_in)
    return out


# --------------------------------------------------------------------------------------
# a8: construct_triplet
# --------------------------------------------------------------------------------------

def duration_overlap(d1, d2):
    """Closed-interval intersection table (utils/utils_func.py:347-373, broadcast form)."""
    s = torch.maximum(d1[:, None, 0], d2[None, :, 0])
    e = torch.minimum(d1[:, None, 1], d2[None, :, 1])
    return torch.stack([s, e], -1), s <= e


def topk_lowest_index(probs, topk):
    """top-k with the HIP path's documented tie rule: among exactly equal probabilities the lowest class id comes first
    (``torch.topk`` leaves that order unspecified; tests/golden/bigc_ties.npz records what it does)."""
    order = np.stack([np.lexsort((np.arange(row.shape[0]), -row)) for row in probs.numpy()])[:, :topk]
    top_c = torch.from_numpy(order.astype(np.int64))
    return torch.gather(probs, 1, top_c), top_c


def construct_triplet(proposal, pred_logits, att, topk, topk_fn=None):
    """models/model_0v10.py:707-785. Returns None or (quintuples i64 (U,5), scores f32 (U,3),
    dura_inters i64 (U,2), query_ids i64 (U,)); rows in the lexicographic order of
    ``torch.unique(dim=0)``, group representative = first candidate with the largest
    predicate probability.  ``topk_fn`` replaces ``torch.topk`` (tests of the tie rule)."""
    nq = pred_logits.shape[0]
    probs = torch.softmax(pred_logits, dim=-1)
    top_p, top_c = torch.topk(probs, topk, dim=-1) if topk_fn is None else topk_fn(probs, topk)
    cand_score = top_p.reshape(-1)
    cand_cat = top_c.reshape(-1)
    cand_query = torch.arange(nq).repeat_interleave(topk)
    so = torch.argmax(att, dim=-1).t().repeat_interleave(topk, dim=0)      # (nq*k, 2)
    duras = proposal.traj_durations
    inter, ok = duration_overlap(duras, duras)
    ok = ok.clone()
    ok.fill_diagonal_(False)
    keep = ok[so[:, 0], so[:, 1]]
    if int(keep.sum()) == 0:
        return None
    so, cand_score, cand_cat, cand_query = so[keep], cand_score[keep], cand_cat[keep], cand_query[keep]
    so_cat = proposal.cat_ids[so]
    so_score = proposal.scores[so]
    quint = torch.cat([cand_cat[:, None], so_cat, so], -1).numpy()         # (P,5)
    trip_scores = torch.cat([cand_score[:, None], so_score], -1)
    order = np.lexsort(tuple(quint[:, c] for c in range(4, -1, -1)))        # stable, col 0 most significant
    sq = quint[order]
    new_group = np.ones(len(order), dtype=bool)
    new_group[1:] = np.any(sq[1:] != sq[:-1], axis=1)
    starts = np.flatnonzero(new_group)
    ends = np.append(starts[1:], len(order))
    rep = []
    sc = cand_score.numpy()
    for s, e in zip(starts, ends):
        members = np.sort(order[s:e])                      # ascending candidate index
        rep.append(members[int(np.argmax(sc[members]))])   # first max
    rep = torch.from_numpy(np.asarray(rep, dtype=np.int64))
    uq = torch.from_numpy(sq[starts].astype(np.int64))
    uscores = trip_scores[rep]
    uquery = cand_query[rep]
    udura = inter[uq[:, 3], uq[:, 4], :]
    fg = uq[:, 0] != 0
    return uq[fg], uscores[fg], udura[fg], uquery[fg]


def forward_test(sd, cfg, proposal_list, topk=10, variant="vidvrd"):
    """models/model_0v10.py:369-388: per-video loop, ``None`` for empty videos."""
    results = []
    for p in proposal_list:
        if p.num_proposals == 0:
            results.append(None)
            continue
        o = encode2decode(sd, cfg, p, variant)
        results.append(construct_triplet(p, o["pred_logits"], o["att"], topk))
    return results
