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


def assemble_features(plan, track_res, dim_boxfeature, i3d_res=None, dim_i3d=0, device="cuda:0"):
    """Device side: the (n_rows, dim_boxfeature + dim_i3d) float32 feature rows of the kept tracklets."""
    lib = _lib.load()
    raw = torch.from_numpy(np.ascontiguousarray(np.asarray(track_res, dtype=np.float32))).to(device)     # one H2D copy of the file
    raw2 = None
    if i3d_res is not None and dim_i3d:
        raw2 = torch.from_numpy(np.ascontiguousarray(np.asarray(i3d_res, dtype=np.float32))).to(device)
        assert raw2.shape == (raw.shape[0], 12 + dim_i3d)
    t = {k: torch.from_numpy(plan[k]).to(device) for k in ("src_a", "src_b", "step_i", "step_div")}
    n_rows = int(plan["src_a"].shape[0])
    out = torch.empty((n_rows, dim_boxfeature + dim_i3d), dtype=torch.float32, device=device)
    _lib.check(lib.bigc_assemble_rows_f32(_lib.current_stream(), raw.data_ptr(), raw.shape[1], 12, dim_boxfeature,
                                          _lib.ptr(raw2), raw2.shape[1] if raw2 is not None else 0, 12, dim_i3d if raw2 is not None else 0,
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
