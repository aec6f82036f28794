"""Result conversion and the relation-detection metrics (mAP, Recall@50/100, tagging P@1/5/10).

Host-side Python like the reference's: ``to_eval_format`` follows
``EvalFmtCvtor.to_eval_format_pr`` (utils/evaluate.py:72-151, ``traj_cutoff`` utils/utils_func.py:523-536) and
``evaluate`` follows ``VidVRDhelperEvalAPIs/visual_relation_detection.py:10-120`` with ``viou`` / ``voc_ap`` of
``VidVRDhelperEvalAPIs/common.py:4-37,65-106``; the per-pair vIoU is vectorised with numpy instead of the
reference's per-frame Python loops.  Pinned by ``tests/golden/evaluator.npz`` (numbers produced by the
reference evaluator on seeded relation sets).
"""
import numpy as np
import torch


_TABLES = None


def name_tables(dataset_type, use_pku=False):
    """(entity id -> name, predicate id -> name) of a dataset, as ``EvalFmtCvtor`` picks them (utils/evaluate.py:13-23,79-82):
    VidVRD proposals of Liu et al. ("PKU") use another category numbering than the standard one.  The vocabularies are data
    (``data/categories.json``, written by tests/golden/make_golden.py from utils/categories_v2.py)."""
    global _TABLES
    if _TABLES is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "categories.json")) as f:
            _TABLES = json.load(f)
    d = dataset_type.lower()
    assert d in ("vidvrd", "vidor"), dataset_type
    if d == "vidvrd":
        return _TABLES["PKU_vidvrd_CatId2name" if use_pku else "vidvrd_CatId2name"], _TABLES["vidvrd_PredId2name"]
    return _TABLES["vidor_CatId2name"], _TABLES["vidor_PredId2name"]


def reset_video_name(video_name, dataset_type):
    """VidOR names are "<group>_<video id>": the evaluator's ground truth is keyed by the id (utils/evaluate.py:26-38)."""
    if dataset_type is not None and dataset_type.lower() == "vidor":
        parts = video_name.split("_")
        assert len(parts) == 2, video_name
        return parts[1]
    return video_name


def to_eval_format(proposal, result, score_mode="mean", enti_names=None, pred_names=None, dataset_type=None, use_pku=False):
    """One video's model output -> ``{video_name: [relation dict, ...]}`` (``EvalFmtCvtor.to_eval_format_pr``,
    utils/evaluate.py:72-151, with ``traj_cutoff`` utils/utils_func.py:523-536; pinned by tests/golden/evalfmt.json).

    ``result`` is ``None`` or the 4-tuple ``BIG_C.forward`` returns (or the 3-tuple ``(quintuples, scores (n,),
    dura_inters)`` the reference's tools pass on).  A relation dict has ``triplet`` (3 names), ``duration``
    ``[start, end)``, ``score``, ``sub_traj`` / ``obj_traj`` (boxes of the subject / object inside the duration).
    Predicate 0 (background) is skipped (:113-114).  ``dataset_type`` ("vidvrd" | "vidor", with ``use_pku`` for the PKU
    proposals' category numbering) selects the dataset's vocabularies; without it (synthetic data) explicit name lists or
    placeholder names are used."""
    if dataset_type is not None:
        enti_names, pred_names = name_tables(dataset_type, use_pku)
    name = reset_video_name(proposal.video_name, dataset_type)
    if result is None:
        return {name: []}
    quint, scores, dura = result[0], result[1], result[2]
    quint = quint.cpu().tolist() if isinstance(quint, torch.Tensor) else quint
    if isinstance(scores, torch.Tensor):
        scores = scores.float().cpu()
        if scores.dim() == 2:                     # tools/eval_vidvrd.py:136: mean of [pred_prob, subj_score, obj_score]
            scores = scores.mean(dim=-1) if score_mode == "mean" else scores[:, 0]
        scores = scores.tolist()
    dura = dura.cpu().tolist() if isinstance(dura, torch.Tensor) else dura
    spans = proposal.traj_durations.cpu().tolist()
    boxes = [b.cpu().numpy() for b in proposal.bboxes_list]
    en = (lambda i: enti_names[i]) if enti_names is not None else (lambda i: "entity_%d" % i)
    pn = (lambda i: pred_names[i]) if pred_names is not None else (lambda i: "predicate_%d" % i)
    out = []
    for (p, sc, oc, st, ot), s, (ds, de) in zip(quint, scores, dura):
        if p == 0:
            continue
        d0, d1 = ds, de + 1                                              # closed -> [start, end)
        assert spans[st][0] <= d0 and d1 <= spans[st][1] + 1 and spans[ot][0] <= d0 and d1 <= spans[ot][1] + 1
        subj = boxes[st][d0 - spans[st][0]: len(boxes[st]) - (spans[st][1] + 1 - d1)]
        obj = boxes[ot][d0 - spans[ot][0]: len(boxes[ot]) - (spans[ot][1] + 1 - d1)]
        assert len(subj) == len(obj) == d1 - d0
        out.append({"triplet": [en(sc), pn(p), en(oc)], "duration": (d0, d1), "score": float(s),
                    "sub_traj": subj.tolist(), "obj_traj": obj.tolist()})
    return {name: out}


def gt_to_eval_format(gt_graph, enti_names=None, pred_names=None, dataset_type=None):
    """GT relations of one video in the evaluator's format (the reference reads them from prepared JSON files,
    VidVRDhelperEvalAPIs/visual_relation_detection.py:250-265): one relation per GT predicate, trajectories cut
    to the predicate's (closed) duration."""
    if dataset_type is not None:
        enti_names, pred_names = name_tables(dataset_type, False)      # ground truth always carries the standard numbering
    en = (lambda i: enti_names[i]) if enti_names is not None else (lambda i: "entity_%d" % i)
    pn = (lambda i: pred_names[i]) if pred_names is not None else (lambda i: "predicate_%d" % i)
    so = gt_graph.adj_matrix.argmax(dim=-1).t().tolist()
    spans = gt_graph.traj_durations.tolist()
    out = []
    for k, (s, o) in enumerate(so):
        d0, d1 = int(gt_graph.pred_durations[k, 0]), int(gt_graph.pred_durations[k, 1]) + 1
        bs, bo = gt_graph.traj_bboxes[s].numpy(), gt_graph.traj_bboxes[o].numpy()
        out.append({"triplet": [en(int(gt_graph.traj_cat_ids[s])), pn(int(gt_graph.pred_cat_ids[k])), en(int(gt_graph.traj_cat_ids[o]))],
                    "duration": (d0, d1),
                    "sub_traj": bs[d0 - spans[s][0]: d1 - spans[s][0]].tolist(), "obj_traj": bo[d0 - spans[o][0]: d1 - spans[o][0]].tolist()})
    return out


def viou(traj_1, duration_1, traj_2, duration_2):
    """Voluminal IoU of two box trajectories given with their ``[start, end)`` durations (common.py:65-106)."""
    s, e = max(duration_1[0], duration_2[0]), min(duration_1[1], duration_2[1])
    if s >= e:
        return 0.0
    a = np.asarray(traj_1, dtype=np.float64)
    b = np.asarray(traj_2, dtype=np.float64)
    aa = a[s - duration_1[0]: e - duration_1[0]]
    bb = b[s - duration_2[0]: e - duration_2[0]]
    w = np.maximum(0, np.minimum(aa[:, 2], bb[:, 2]) - np.maximum(aa[:, 0], bb[:, 0]) + 1)
    h = np.maximum(0, np.minimum(aa[:, 3], bb[:, 3]) - np.maximum(aa[:, 1], bb[:, 1]) + 1)
    inter = float((w * h).sum())
    v1 = float(((a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)).sum())
    v2 = float(((b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)).sum())
    return inter / (v1 + v2 - inter)


def voc_ap(rec, prec):
    """Area under the precision envelope (common.py:4-37, the non-07 branch)."""
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def _prec_rec(tp, n_gt):
    fp = ~tp
    cum_tp = np.cumsum(tp).astype(np.float32)
    cum_fp = np.cumsum(fp).astype(np.float32)
    eps = np.finfo(np.float32).eps
    return cum_tp / np.maximum(cum_tp + cum_fp, eps), cum_tp / np.maximum(n_gt, eps)


def viou_tables_device(groundtruth, prediction, device):
    """vIoU of every (prediction, ground-truth relation) pair with EQUAL triplets of every video, subject and object
    trajectories, in two launches of the HIP kernel ``bigc_viou_pairs`` (the reference calls a Python frame loop per pair,
    common.py:65-106 from visual_relation_detection.py:21-27; on VidVRD-test that is the larger part of the evaluation time).
    -> {video: dict (pred index, gt index) -> min(subject vIoU, object vIoU)} for the videos that have both sides."""
    from . import _lib
    lib = _lib.load()
    boxes = {"p": [], "g": []}
    segs = {"ps": [], "po": [], "gs": [], "go": []}
    rows = {"p": 0, "g": 0}
    pairs, owner = [], []

    def add(side, traj, start):
        a = np.asarray(traj, dtype=np.float32).reshape(-1, 4)
        boxes[side].append(a)
        seg = [rows[side], int(start), a.shape[0]]
        rows[side] += a.shape[0]
        return seg

    n_p = n_g = 0
    for vid, gts in groundtruth.items():
        preds = prediction.get(vid, [])
        if not gts or not preds:
            continue
        by_trip = {}
        for k, g in enumerate(gts):
            by_trip.setdefault(tuple(g["triplet"]), []).append(k)
        g_base = n_g
        for g in gts:
            segs["gs"].append(add("g", g["sub_traj"], g["duration"][0]))
            segs["go"].append(add("g", g["obj_traj"], g["duration"][0]))
            n_g += 1
        for i, p in enumerate(preds):
            ks = by_trip.get(tuple(p["triplet"]))
            if not ks:
                continue
            segs["ps"].append(add("p", p["sub_traj"], p["duration"][0]))
            segs["po"].append(add("p", p["obj_traj"], p["duration"][0]))
            for k in ks:
                pairs.append((n_p, g_base + k))
                owner.append((vid, i, k))
            n_p += 1
    tables = {}
    if not pairs:
        return tables
    dev = torch.device(device)
    bp = torch.from_numpy(np.concatenate(boxes["p"])).to(dev)
    bg = torch.from_numpy(np.concatenate(boxes["g"])).to(dev)
    pr = torch.tensor(pairs, dtype=torch.int32, device=dev)
    res = []
    for a, b in (("ps", "gs"), ("po", "go")):
        sa = torch.tensor(segs[a], dtype=torch.int32, device=dev)
        sb = torch.tensor(segs[b], dtype=torch.int32, device=dev)
        out = torch.empty((len(pairs),), dtype=torch.float32, device=dev)
        _lib.check(lib.bigc_viou_pairs(_lib.current_stream(), bp.data_ptr(), sa.data_ptr(), bg.data_ptr(), sb.data_ptr(), pr.data_ptr(),
                                       len(pairs), out.data_ptr()))
        res.append(out)
    ov = torch.minimum(res[0], res[1]).cpu().numpy()
    for (vid, i, k), v in zip(owner, ov):
        tables.setdefault(vid, {})[(i, k)] = float(v)
    return tables


def detection_scores(gt_relations, pred_relations, viou_threshold, table=None):
    """Greedy matching in descending score order: a prediction hits the not-yet-detected GT relation of the same
    triplet with the largest min(subject vIoU, object vIoU) >= threshold (visual_relation_detection.py:10-37).
    ``table`` (from ``viou_tables_device``): precomputed overlaps keyed by (index into pred_relations, gt index); a value
    within 1e-5 of the threshold is recomputed on the host in float64, as the reference computes all of them."""
    order = sorted(range(len(pred_relations)), key=lambda i: pred_relations[i]["score"], reverse=True)
    detected = np.zeros(len(gt_relations), dtype=bool)
    hit = np.full(len(order), -np.inf)
    gt_trip = [tuple(g["triplet"]) for g in gt_relations]
    for rank, i in enumerate(order):
        p = pred_relations[i]
        best, k_best = -np.inf, -1
        t = tuple(p["triplet"])
        for k, g in enumerate(gt_relations):
            if detected[k] or gt_trip[k] != t:
                continue
            ov = table.get((i, k)) if table is not None else None
            if ov is None or abs(ov - viou_threshold) < 1e-5:
                ov = min(viou(p["sub_traj"], p["duration"], g["sub_traj"], g["duration"]),
                         viou(p["obj_traj"], p["duration"], g["obj_traj"], g["duration"]))
            if ov >= viou_threshold and ov > best:
                best, k_best = ov, k
        if k_best >= 0:
            hit[rank] = p["score"]
            detected[k_best] = True
    prec, rec = _prec_rec(np.isfinite(hit), len(gt_relations))
    return prec, rec, hit


def tagging_scores(gt_relations, pred_relations):
    """(visual_relation_detection.py:40-61): unique predicted triplets in score order vs the set of GT triplets."""
    preds = sorted(pred_relations, key=lambda x: x["score"], reverse=True)
    gt = set(tuple(r["triplet"]) for r in gt_relations)
    seen, scores = [], []
    for r in preds:
        t = tuple(r["triplet"])
        if t not in seen:
            seen.append(t)
            scores.append(r["score"])
    scores = np.asarray(scores, dtype=np.float64)
    for i, t in enumerate(seen):
        if t not in gt:
            scores[i] = -np.inf
    prec, rec = _prec_rec(np.isfinite(scores), len(gt))
    return prec, rec, scores


def evaluate(groundtruth, prediction, viou_threshold=0.5, det_nreturns=(50, 100), tag_nreturns=(1, 5, 10), device=None):
    """-> (mean_ap, {n: recall@n}, {n: tagging precision@n})   (visual_relation_detection.py:64-120).
    With ``device`` (a CUDA device) the vIoU overlaps come from one batched HIP kernel instead of per-pair host loops."""
    tables = viou_tables_device(groundtruth, prediction, device) if device is not None else {}
    video_ap, tot_scores, tot_tp = {}, {n: [] for n in det_nreturns}, {n: [] for n in det_nreturns}
    prec_at_n = {n: [] for n in tag_nreturns}
    tot_gt = 0
    for vid, gt_relations in groundtruth.items():
        if len(gt_relations) == 0:
            continue
        tot_gt += len(gt_relations)
        preds = prediction.get(vid, [])
        prec, rec, scores = detection_scores(gt_relations, preds, viou_threshold, tables.get(vid) if device is not None else None)
        video_ap[vid] = voc_ap(rec, prec)
        tp = np.isfinite(scores)
        for n in det_nreturns:
            cut = min(n, scores.size)
            tot_scores[n].append(scores[:cut])
            tot_tp[n].append(tp[:cut])
        tag_prec, _, _ = tagging_scores(gt_relations, preds)
        for n in tag_nreturns:
            cut = min(n, tag_prec.size)
            prec_at_n[n].append(tag_prec[cut - 1] if cut > 0 else 0.)
    mean_ap = float(np.mean(list(video_ap.values())))
    rec_at_n = {}
    for n in det_nreturns:
        scores, tps = np.concatenate(tot_scores[n]), np.concatenate(tot_tp[n])
        tps = tps[np.argsort(scores)[::-1]]
        cum_tp = np.cumsum(tps).astype(np.float32)
        rec = cum_tp / np.maximum(tot_gt, np.finfo(np.float32).eps)
        rec_at_n[n] = float(rec[-1]) if rec.size else 0.0
    return mean_ap, rec_at_n, {n: float(np.mean(v)) for n, v in prec_at_n.items()}
