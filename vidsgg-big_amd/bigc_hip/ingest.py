"""Tracking-result ingest: a video's ``.npy`` of per-box rows -> the tracklet proposal the models consume, with the feature
rows assembled on the GPU (reference: ``Dataset_pku._get_proposal`` /
``Dataset_pku_i3d._get_proposal`` dataloaders/dataloader_vidvrd.py:485-573, :585-700, ``TrajProposal`` :14-52).

Row format (tracking_results/format_demo.py:9-21): ``[fid, tid, x,y,w,h (tracker), conf, cat, x,y,w,h (detector), feat...]``.

Split of work
  * host (numpy, float64 as the reference's Python lists): grouping by tracklet id in order of first appearance,
    box = mean of tracker and detector boxes (:517-521; tracker box alone and confidence 0 when the box has no
    detection, :510-513), majority-vote category (:536), min-frames filter (:548), ``numpy.linspace`` interpolation of the
    5-vectors [box, conf] across gaps (:555-556, utils/utils_func.py:292-317), score = mean confidence over the
    interpolated rows (:62-64), sort by score + truncation (:40-46).  These are n_rows x 5 numbers.
  * device (``bigc_assemble_rows_f32``): the feature rows (n_rows x (dim_box + dim_i3d)), i.e. everything heavy: copies of
    the raw rows, zero rows for undetected boxes, and the gap rows interpolated with ``numpy.linspace``'s exact float64
    arithmetic (including its any-step-zero branch) before the cast to float32 -- bit-identical to the reference.
"""
import numpy as np
import torch

from . import _lib
from .synthetic import TrajProposal


def plan_tracklets(track_res, dim_boxfeature, min_frames_th, max_proposal):
    """Host bookkeeping.  -> dict with per kept tracklet (final order): cat_ids, scores, durations (closed), boxes (list of
    (L,4) float32) and the row assembly tables src_a, src_b, step_i, step_div (int32, one entry per output row)."""
    raw = np.asarray(track_res)
    assert raw.ndim == 2 and raw.shape[1] == 12 + dim_boxfeature, "len(box_info)=={}".format(raw.shape[1] if raw.ndim == 2 else "ragged")
    head = raw[:, :12].astype(np.float64)
    fid, tid, cat = head[:, 0].astype(np.int64), head[:, 1].astype(np.int64), head[:, 7].astype(np.int64)
    # tracklets in order of first appearance (dict insertion order, :490)
    _, first = np.unique(tid, return_index=True)
    order = tid[np.sort(first)]
    detected = cat > 0
    tb = np.stack([head[:, 2], head[:, 3], head[:, 2] + head[:, 4], head[:, 3] + head[:, 5]], 1)          # tracker xyxy
    db = np.stack([head[:, 8], head[:, 9], head[:, 8] + head[:, 10], head[:, 9] + head[:, 11]], 1)        # detector xyxy
    box5 = np.concatenate([np.where(detected[:, None], (db + tb) / 2, tb), np.where(detected, head[:, 6], 0.0)[:, None]], 1)
    kept = []
    for t in order:
        rows = np.nonzero(tid == t)[0]                         # file order
        cats = cat[rows][detected[rows]]
        if len(cats) == 0:
            continue                                           # never detected: background (:532-533)
        category = int(np.argmax(np.bincount(cats)))
        if len(rows) < min_frames_th or category == 0:
            continue
        f = fid[rows]
        start, end = int(f.min()), int(f.max()) + 1
        b5 = box5[rows]
        a = np.where(detected[rows], rows, -1).astype(np.int32)   # source row of each present frame (-1: zero features)
        src_a, src_b, step_i, step_div = [a], [], [], []
        if len(rows) != end - start:                           # gaps -> interpolate (:555-556)
            cut = np.nonzero(np.diff(f) > 1)[0] + 1
            seg_f, seg_b, seg_a = np.split(f, cut), np.split(b5, cut, axis=0), np.split(a, cut)
            parts_b, src_a, src_b, step_i, step_div = [], [], [], [], []
            for i in range(1, len(seg_f)):
                left, right = int(seg_f[i - 1][-1]), int(seg_f[i][0])
                n_in = right - left - 1
                parts_b += [seg_b[i - 1], np.linspace(seg_b[i - 1][-1], seg_b[i][0], num=n_in + 2, axis=0)[1:-1]]
                src_a += [seg_a[i - 1], np.full(n_in, seg_a[i - 1][-1], np.int32)]
                src_b += [np.full(len(seg_a[i - 1]), -2, np.int32), np.full(n_in, seg_a[i][0], np.int32)]
                step_i += [np.zeros(len(seg_a[i - 1]), np.int32), np.arange(1, n_in + 1, dtype=np.int32)]
                step_div += [np.ones(len(seg_a[i - 1]), np.int32), np.full(n_in, n_in + 1, np.int32)]
            parts_b.append(seg_b[-1])
            src_a.append(seg_a[-1])
            n_last = len(seg_a[-1])
            src_b.append(np.full(n_last, -2, np.int32)); step_i.append(np.zeros(n_last, np.int32)); step_div.append(np.ones(n_last, np.int32))
            b5 = np.concatenate(parts_b, 0)
        else:
            n = len(rows)
            src_b, step_i, step_div = [np.full(n, -2, np.int32)], [np.zeros(n, np.int32)], [np.ones(n, np.int32)]
        assert len(b5) == end - start
        score = sum(float(x) for x in b5[:, 4]) / len(b5)       # Python-float running sum as the reference (:62-64)
        kept.append(dict(cat=category, score=score, dura=(start, end - 1), boxes=b5[:, :4],
                         src_a=np.concatenate(src_a), src_b=np.concatenate(src_b), step_i=np.concatenate(step_i),
                         step_div=np.concatenate(step_div)))
    if not kept:
        return None
    scores = torch.FloatTensor([k["score"] for k in kept])
    index = torch.argsort(scores, descending=True)[:max_proposal].tolist()        # (:40-42)
    kept = [kept[i] for i in index]
    return dict(cat_ids=[k["cat"] for k in kept], scores=scores[index], durations=[k["dura"] for k in kept],
                boxes=[torch.FloatTensor(k["boxes"]) for k in kept],
                **{key: np.concatenate([k[key] for k in kept]).astype(np.int32) for key in ("src_a", "src_b", "step_i", "step_div")})


def assemble_features(plan, track_res, dim_boxfeature, i3d_res=None, dim_i3d=0, device="cuda:0", col2=12):
    """Device side: the (n_rows, dim_boxfeature + dim_i3d) float32 feature rows of the kept tracklets.  ``i3d_res``: the second per-box
    file whose columns [col2, col2 + dim_i3d) ride behind the box features (VidVRD: the I3D rows, 12 head columns; VidOR: the classeme
    rows, no head columns)."""
    lib = _lib.load()
    raw = torch.from_numpy(np.ascontiguousarray(np.asarray(track_res, dtype=np.float32))).to(device)     # one H2D copy of the file
    raw2 = None
    if i3d_res is not None and dim_i3d:
        raw2 = torch.from_numpy(np.ascontiguousarray(np.asarray(i3d_res, dtype=np.float32))).to(device)
        assert raw2.shape == (raw.shape[0], col2 + dim_i3d)
    t = {k: torch.from_numpy(plan[k]).to(device) for k in ("src_a", "src_b", "step_i", "step_div")}
    n_rows = int(plan["src_a"].shape[0])
    out = torch.empty((n_rows, dim_boxfeature + dim_i3d), dtype=torch.float32, device=device)
    _lib.check(lib.bigc_assemble_rows_f32(_lib.current_stream(), raw.data_ptr(), raw.shape[1], 12, dim_boxfeature,
                                          _lib.ptr(raw2), raw2.shape[1] if raw2 is not None else 0, col2, dim_i3d if raw2 is not None else 0,
                                          t["src_a"].data_ptr(), t["src_b"].data_ptr(), t["step_i"].data_ptr(), t["step_div"].data_ptr(),
                                          out.data_ptr(), out.shape[1], n_rows))
    return out


def load_pku_tracklets(video_name, track_res, dim_boxfeature, min_frames_th, max_proposal, i3d_res=None, dim_i3d=0, device="cuda:0",
                       video_len=None, video_wh=(1280, 720)):
    """-> TrajProposal with device-resident features (views of one assembled tensor), or an empty proposal."""
    plan = plan_tracklets(track_res, dim_boxfeature, min_frames_th, max_proposal)
    if plan is None:
        return TrajProposal(video_name, video_len or 0, video_wh, [], [], [], [], [])
    feats = assemble_features(plan, track_res, dim_boxfeature, i3d_res, dim_i3d, device)
    lens = [int(b.shape[0]) for b in plan["boxes"]]
    # the tracklets stay views of the one assembled feature tensor (and of one box tensor): the batch packer reads their addresses
    # off the block (TrajProposal.from_blocks)
    boxes = torch.cat(plan["boxes"], 0).to(feats.device)
    return TrajProposal.from_blocks(video_name, video_len or (max(d[1] for d in plan["durations"]) + 1), video_wh, plan["cat_ids"], plan["scores"],
                                    plan["durations"], lens, boxes, feats)


# ---------------------------------------------------------------------------------------------------------------------------------------
# VidOR (dataloaders/dataloader_vidor_v3.py:389-485 ``_get_proposal``, :19-69 ``TrajProposal``, :265-277 ``load_video_features``)
# ---------------------------------------------------------------------------------------------------------------------------------------

def plan_vidor_tracklets(track_res, dim_boxfeature, min_frames_th, max_proposal, score_th):
    """Host bookkeeping of the VidOR tracking results.  ``track_res``: per-box rows ``[fid, tid, x, y, w, h (tracker), conf, cat,
    x, y, w, h (detector), feat...]`` of length ``12 + dim_boxfeature``, or of length 6 (a tracker-only box: skipped, :408-409, so its
    frame becomes a gap) -- a 2-D array, or the reference's object array / list of ragged rows.  Differences from the VidVRD loader
    (``plan_tracklets``): the box is the TRACKER box with the detector's confidence (:411-418, no averaging with the detector box),
    every full row carries features, the classeme row of a box is picked by its index in the FILE (:428), a tracklet without full rows
    or shorter than ``min_frames_th`` is background (:431-445), and the proposals whose mean confidence is not above ``score_th`` go
    BEFORE the top-``max_proposal`` cut (:39-53).
    -> (plan as ``plan_tracklets``' with ``file_rows``: the file index of every dense row, or None when nothing is left;
        dense (n_full, 12 + dim_boxfeature) float32 rows)"""
    if isinstance(track_res, np.ndarray) and track_res.dtype != object and track_res.ndim == 2:
        assert track_res.shape[1] in (6, 12 + dim_boxfeature), "len(box_info)=={}".format(track_res.shape[1])
        if track_res.shape[1] == 6:
            return None, None
        file_rows = np.arange(track_res.shape[0])
        dense = np.ascontiguousarray(track_res, dtype=np.float32)
        tid_all = track_res[:, 1].astype(np.int64)
    else:
        rows = [r if isinstance(r, list) else np.asarray(r).tolist() for r in track_res]
        for r in rows:
            assert len(r) == 6 or len(r) == 12 + dim_boxfeature, "len(box_info)=={}".format(len(r))
        file_rows = np.asarray([i for i, r in enumerate(rows) if len(r) != 6], dtype=np.int64)
        tid_all = np.asarray([int(r[1]) for r in rows], dtype=np.int64)
        if len(file_rows) == 0:
            return None, None
        dense = np.asarray([rows[i] for i in file_rows], dtype=np.float32)
    head = dense[:, :12].astype(np.float64)
    fid, tid, cat = head[:, 0].astype(np.int64), head[:, 1].astype(np.int64), head[:, 7].astype(np.int64)
    _, first = np.unique(tid_all, return_index=True)             # tracklets in order of first appearance over ALL rows (:395)
    order = tid_all[np.sort(first)]
    box5 = np.stack([head[:, 2], head[:, 3], head[:, 2] + head[:, 4], head[:, 3] + head[:, 5], head[:, 6]], 1)
    kept = []
    for t in order:
        rows_t = np.nonzero(tid == t)[0]                        # dense rows of the tracklet, file order
        if len(rows_t) == 0:
            continue                                           # only tracker-only boxes: category 0 (:431-432)
        category = int(np.argmax(np.bincount(cat[rows_t])))    # (:435)
        if len(rows_t) < min_frames_th or category == 0:
            continue
        f = fid[rows_t]
        start, end = int(f.min()), int(f.max()) + 1
        b5 = box5[rows_t]
        a = rows_t.astype(np.int32)
        if len(rows_t) != end - start:                         # gaps -> interpolate (:458-460, utils/utils_func.py:292-317)
            cut = np.nonzero(np.diff(f) > 1)[0] + 1
            seg_f, seg_b, seg_a = np.split(f, cut), np.split(b5, cut, axis=0), np.split(a, cut)
            parts_b, src_a, src_b, step_i, step_div = [], [], [], [], []
            for i in range(1, len(seg_f)):
                left, right = int(seg_f[i - 1][-1]), int(seg_f[i][0])
                n_in = right - left - 1
                parts_b += [seg_b[i - 1], np.linspace(seg_b[i - 1][-1], seg_b[i][0], num=n_in + 2, axis=0)[1:-1]]
                src_a += [seg_a[i - 1], np.full(n_in, seg_a[i - 1][-1], np.int32)]
                src_b += [np.full(len(seg_a[i - 1]), -2, np.int32), np.full(n_in, seg_a[i][0], np.int32)]
                step_i += [np.zeros(len(seg_a[i - 1]), np.int32), np.arange(1, n_in + 1, dtype=np.int32)]
                step_div += [np.ones(len(seg_a[i - 1]), np.int32), np.full(n_in, n_in + 1, np.int32)]
            parts_b.append(seg_b[-1])
            src_a.append(seg_a[-1])
            n_last = len(seg_a[-1])
            src_b.append(np.full(n_last, -2, np.int32)); step_i.append(np.zeros(n_last, np.int32)); step_div.append(np.ones(n_last, np.int32))
            b5 = np.concatenate(parts_b, 0)
        else:
            n = len(rows_t)
            src_a, src_b, step_i, step_div = [a], [np.full(n, -2, np.int32)], [np.zeros(n, np.int32)], [np.ones(n, np.int32)]
        assert len(b5) == end - start
        score = sum(float(x) for x in b5[:, 4]) / len(b5)       # Python-float running sum as the reference (:79-81)
        kept.append(dict(cat=category, score=score, dura=(start, end - 1), boxes=b5[:, :4],
                         src_a=np.concatenate(src_a), src_b=np.concatenate(src_b), step_i=np.concatenate(step_i),
                         step_div=np.concatenate(step_div)))
    if not kept:
        return None, None
    scores = torch.FloatTensor([k["score"] for k in kept])
    above = torch.where(scores > score_th)[0]                    # score clipping first (:39-50) ...
    if len(above) == 0:
        return None, None
    scores = scores[above]
    order2 = torch.argsort(scores, descending=True)[:max_proposal]       # ... then the top-N cut (:53-55)
    kept = [kept[int(above[i])] for i in order2.tolist()]
    plan = dict(cat_ids=[k["cat"] for k in kept], scores=scores[order2], durations=[k["dura"] for k in kept],
                boxes=[torch.FloatTensor(k["boxes"]) for k in kept], file_rows=file_rows,
                **{key: np.concatenate([k[key] for k in kept]).astype(np.int32) for key in ("src_a", "src_b", "step_i", "step_div")})
    return plan, dense


def load_vidor_tracklets(video_name, track_res, classeme, dim_boxfeature, min_frames_th, max_proposal, score_th, device="cuda:0",
                         video_len=None, video_wh=(1280, 720)):
    """``Dataset._get_proposal`` of the VidOR loader (dataloaders/dataloader_vidor_v3.py:389-485) + its ``TrajProposal`` (:19-69):
    -> TrajProposal whose feature rows ``[RoI (dim_boxfeature) | classeme (300)]`` were assembled on the device (views of one tensor), or
    an empty proposal.  ``classeme``: the ``<video>_clsme.npy`` array, one row per row of ``track_res``."""
    plan, dense = plan_vidor_tracklets(track_res, dim_boxfeature, min_frames_th, max_proposal, score_th)
    if plan is None:
        return TrajProposal(video_name, video_len or 0, video_wh, [], [], [], [], [])
    cls = np.ascontiguousarray(np.asarray(classeme, dtype=np.float32)[plan["file_rows"]])          # the full rows' classemes, dense order
    feats = assemble_features(plan, dense, dim_boxfeature, cls, cls.shape[1], device, col2=0)
    lens = [int(b.shape[0]) for b in plan["boxes"]]
    boxes = torch.cat(plan["boxes"], 0).to(feats.device)
    return TrajProposal.from_blocks(video_name, video_len or (max(d[1] for d in plan["durations"]) + 1), video_wh, plan["cat_ids"], plan["scores"],
                                    plan["durations"], lens, boxes, feats)


def load_video_features(video_feature_dir, video_names, device=None):
    """``Dataset.load_video_features`` (dataloaders/dataloader_vidor_v3.py:265-277): {video_name: (num_clips, 1024) clip features} for
    the grounding stage, optionally moved to ``device``."""
    import os
    out = {}
    for name in video_names:
        f = torch.from_numpy(np.load(os.path.join(video_feature_dir, name + ".npy")))
        out[name] = f.to(device) if device is not None else f
    return out


# ---------------------------------------------------------------------------------------------------------------------------------------
# The on-disk cache users of the reference already have (dataloaders/dataloader_vidvrd.py:170-183, dataloader_vidor_v3.py:229-246):
# ``datasets/cache/<tag>_VidVRD<split>_th_<m>-<p>-<q>.pkl`` = pickle of {video_name: (TrajProposal, VideoGraph)} holding INSTANCES of the
# reference's classes.  Read here without the reference on PYTHONPATH: a restricted unpickler maps those class names to plain holders
# (pickle restores an instance's __dict__ without calling __init__) and nothing else but torch / numpy / builtins is allowed to load.
# ---------------------------------------------------------------------------------------------------------------------------------------

class _Holder(object):
    """plain attribute bag a pickled reference object is restored into"""


def _restricted_unpickler(f):
    import pickle

    class U(pickle.Unpickler):
        def find_class(self, module, name):
            if name in ("TrajProposal", "VideoGraph") and (module.startswith("dataloaders.") or module in ("__main__", "dataloader_vidvrd", "dataloader_vidor", "dataloader_vidor_v3")):
                return _Holder
            root = module.split(".")[0]
            if root in ("torch", "numpy", "collections", "_codecs") or (root == "builtins" and name in ("dict", "list", "tuple", "set", "frozenset", "int", "float", "str", "bytes", "bool", "complex", "slice", "range", "bytearray", "object")):
                return super().find_class(module, name)
            raise pickle.UnpicklingError("the cache holds a %s.%s object: only the reference's TrajProposal / VideoGraph, tensors and arrays are read" % (module, name))
    return U(f)


def proposal_from_reference(h, device=None):
    """a restored reference ``TrajProposal`` (dataloaders/dataloader_vidvrd.py:14-77, dataloader_vidor_v3.py:19-98) -> this package's
    block-backed container (one box block + one feature block per video; ``TrajProposal.from_blocks``), optionally on ``device``"""
    name = getattr(h, "video_name", None)
    wh = tuple(getattr(h, "video_wh", (1280, 720)))
    n = int(getattr(h, "num_proposals", 0))
    if n == 0:
        return TrajProposal(name, int(getattr(h, "video_len", 0) or 0), wh, [], [], [], [], [])
    boxes = [torch.as_tensor(b, dtype=torch.float32) for b in h.bboxes_list[:n]]
    feats = [torch.as_tensor(f, dtype=torch.float32) for f in h.features_list[:n]]
    dura = torch.as_tensor(h.traj_durations, dtype=torch.long)[:n]
    lens = [int(b.shape[0]) for b in boxes]
    assert all(int(f.shape[0]) == l for f, l in zip(feats, lens)) and all(int(dura[i, 1] - dura[i, 0] + 1) == lens[i] for i in range(n))
    bb, fb = torch.cat(boxes, 0).contiguous(), torch.cat(feats, 0).contiguous()
    if device is not None:
        bb, fb = bb.to(device), fb.to(device)
    video_len = getattr(h, "video_len", None)
    return TrajProposal.from_blocks(name, int(video_len) if video_len is not None else int(dura[:, 1].max()) + 1, wh, torch.as_tensor(h.cat_ids)[:n],
                                    torch.as_tensor(h.scores)[:n], dura, lens, bb, fb)


def graph_from_reference(h):
    """a restored reference ``VideoGraph`` (dataloaders/dataloader_vidvrd.py:84-140) -> ``bigc_hip.synthetic.VideoGraph`` (same fields, already
    in closed-interval form as the reference's constructor left them)"""
    from .synthetic import VideoGraph
    return VideoGraph(h.video_name, h.video_len, h.video_wh, h.traj_cat_ids, h.traj_durations, h.traj_bboxes, h.pred_cat_ids, h.pred_durations, h.adj_matrix)


def load_reference_cache(path, device=None, with_gt=True):
    """``datasets/cache/*.pkl`` of the reference -> {video_name: (TrajProposal, VideoGraph or None)} in this package's containers.  VidOR caches
    made with video features hold 3-tuples (video_feature, proposal, graph) per video: returned as (proposal, graph, video_feature)."""
    with open(path, "rb") as f:
        data = _restricted_unpickler(f).load()
    out = {}
    for name, item in data.items():
        item = tuple(item) if isinstance(item, (tuple, list)) else (item,)
        props = [x for x in item if isinstance(x, _Holder) and hasattr(x, "bboxes_list") or (isinstance(x, _Holder) and getattr(x, "num_proposals", None) == 0)]
        graphs = [x for x in item if isinstance(x, _Holder) and hasattr(x, "adj_matrix")]
        feats = [x for x in item if isinstance(x, torch.Tensor)]
        assert len(props) == 1, "video %s: expected one TrajProposal in the cache entry" % name
        p = proposal_from_reference(props[0], device)
        g = graph_from_reference(graphs[0]) if (with_gt and graphs) else None
        out[name] = (p, g) + ((feats[0],) if feats else ())
    return out
