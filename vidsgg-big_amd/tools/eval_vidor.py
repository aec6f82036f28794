#!/usr/bin/env python3
"""VidOR inference + evaluation on the HIP path with the flags of the reference's ``tools/eval_vidor.py:285-300``:
``--eval_cls_only`` runs the BIG-C classification stage (``evaluate_cls_stage`` :21-138) and saves the per-video results
pickle; without it the grounding stage re-scores those results (``evaluate_combined`` :141-282, needs
``--cls_stage_result_path``); ``--json_results_path`` only evaluates a saved prediction file.
``--synthetic N`` generates the videos, clip features and ground truth instead of reading a dataset.

Several GPUs: start it under ``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1``: one process per GPU, each
scoring its size-balanced share of the videos; per-video results are gathered on rank 0, which saves and evaluates them (SURVEY.md
section 8e; replaces utils/DataParallel.py:45-113 around tools/eval_vidor.py:218-256)."""
import argparse
import json
import os
import pickle
import time

import torch

from _common import (ensure_grounding_files, ensure_table_files, init_process, load_checkpoint_into, parse_config_py, shard_for_inference,
                     synthetic_vidor_index, dataset_index)

from bigc_hip import evaluate as E
from bigc_hip.pipeline import combine_with_grounding
from bigc_hip.shard import gather_results
from models import BIG_C_vidor, Base_C, DEBUG


RAW = {}        # --tracking_dir / --classeme_dir / --video_feature_dir / --video_info_path / --gt_json_path of this run (set in __main__)


def _load_data(cfg, synthetic, n_trajs, clip_dim, device=None):
    """the data set as an index of (proposal, gt_graph, clip features) items, loaded per rank on demand (_common.VideoIndex)"""
    if synthetic:
        return synthetic_vidor_index(cfg["model_config"] if not clip_dim else {}, synthetic, n_trajs, seed=2468, clip_dim=clip_dim)
    if RAW.get("tracking_dir"):
        # raw tracking results: one `<video>.npy` of per-box rows (length 12 + dim_boxfeature, or 6 for tracker-only boxes) and one
        # `<video>_clsme.npy` of classeme rows per video, ingested here (bigc_hip/ingest.py: dataloaders/dataloader_vidor_v3.py:389-485
        # restated, the feature rows assembled on the GPU); the grounding stage's clip features from --video_feature_dir (:265-277)
        import glob
        import numpy as np
        from _common import VideoIndex
        from bigc_hip.ingest import load_vidor_tracklets
        ds_cfg = cfg["test_dataset_config"]
        paths = sorted(glob.glob(os.path.join(RAW["tracking_dir"], "*.npy")))
        names = [os.path.splitext(os.path.basename(path))[0] for path in paths]
        info = json.load(open(RAW["video_info_path"])) if RAW.get("video_info_path") else {}

        def load(i):
            nm = names[i]
            vi = info.get(nm, {})
            p = load_vidor_tracklets(nm, np.load(paths[i], allow_pickle=True), np.load(os.path.join(RAW["classeme_dir"], nm + "_clsme.npy")),
                                     ds_cfg["dim_boxfeature"], ds_cfg["min_frames_th"], ds_cfg["max_proposal"], ds_cfg["score_th"],
                                     device=device or "cuda:0", video_len=vi.get("video_len"), video_wh=tuple(vi.get("video_wh", (1280, 720))))
            clip = None
            if clip_dim and RAW.get("video_feature_dir"):
                clip = torch.from_numpy(np.load(os.path.join(RAW["video_feature_dir"], nm + ".npy")))       # (num_clips, 1024)
            return (p, None, clip)
        return VideoIndex([max(1, os.path.getsize(path)) for path in paths], load, names)
    from dataloaders.dataloader_vidor_v3 import Dataset                 # the reference's loader (tools/eval_vidor.py:11)
    ds = Dataset(**cfg["test_dataset_config"])
    return dataset_index(ds, lambda it: (it[1], it[2], it[0]) if len(it) == 3 else (it[0], it[1], None))


def _report(groundtruth, predictions, output_dir, name, save_json, device=None):
    if save_json:
        with open(os.path.join(output_dir, name), "w") as f:
            json.dump(predictions, f)
    if not groundtruth:
        with open(RAW.get("gt_json_path") or "datasets/GT_json_for_eval/VidORval_gts.json") as f:      # VidVRDhelperEvalAPIs/visual_relation_detection.py:250-256
            groundtruth = json.load(f)
    mean_ap, rec, prec = E.evaluate(groundtruth, predictions, device=device)
    print("detection mean AP (used in challenge): {}".format(mean_ap))
    print("detection recall: {}".format(rec))
    print("tagging precision: {}".format(prec))
    return mean_ap, rec, prec


def evaluate_cls_stage(cfg_path, ckpt_path=None, save_tag="", output_dir=None, cuda=0, save_json=False, synthetic=0, n_trajs=32,
                       model_class=BIG_C_vidor):
    cfg = parse_config_py(cfg_path)
    topk = cfg["inference_config"]["topk"]
    save_tag = "topk{}_{}".format(topk, save_tag)                       # tools/eval_vidor.py:47
    output_dir = output_dir or os.path.dirname(os.path.abspath(cfg_path))
    os.makedirs(output_dir, exist_ok=True)
    rank, world, device, dist = init_process(cuda)
    torch.manual_seed(0)            # without --ckpt_path (synthetic smoke runs) every rank must draw the same initial weights
    model = model_class(ensure_table_files(cfg["model_config"], output_dir), is_train=False).to(device)
    if ckpt_path:
        load_checkpoint_into(model, ckpt_path, device)
    model.eval()
    data = _load_data(cfg, synthetic, n_trajs, None, device)
    predictions, groundtruth, raw = {}, {}, {}
    mine, my_data = shard_for_inference(data, data.costs, rank, world)          # only this rank's videos are loaded
    t0 = time.time()
    my_results = []
    # one video per forward as the reference (tools/eval_vidor.py:91-121), two forwards in flight where the model offers it (BIG-C:
    # forward_async -- the host side of video i + 1 runs while the GPU works on video i; Base_C scores synchronously)
    asynchronous = hasattr(model, "forward_async")
    from collections import deque
    inflight = deque()

    def finish(vi, head):
        p, g, _ = my_data[vi]
        r = (head.result() if asynchronous else head)[0]
        r = None if r is None else tuple(t.cpu() for t in r)
        # VidOR vocabulary, video name reset to the id the ground-truth JSON is keyed by (utils/evaluate.py:26-38); converted by the
        # rank that holds the video's tracklets
        pred = E.to_eval_format(p.to(torch.device("cpu")), r, dataset_type="vidor")
        # (real data is evaluated against datasets/GT_json_for_eval/VidORval_gts.json, _report)
        gt = (E.reset_video_name(p.video_name, "vidor"), E.gt_to_eval_format(g, dataset_type="vidor")) if (g is not None and synthetic) else None
        my_results.append((p.video_name, r, pred, gt))

    for vi in range(len(my_data)):
        pv = my_data[vi][0].to(device)
        with torch.no_grad():
            inflight.append((vi, model.forward_async([pv], topk=topk) if asynchronous else model([pv], topk=topk)))
        if len(inflight) == 2:
            finish(*inflight.popleft())
    while inflight:
        finish(*inflight.popleft())
    torch.cuda.synchronize()
    all_results = gather_results(my_results, mine, len(data), dist)
    if all_results is None:                                 # ranks other than 0 are done
        dist.destroy_process_group()
        return None
    for name, r, pred, gt in all_results:
        raw[name] = None if r is None else list(r)
        predictions.update(pred)
        if gt is not None:
            groundtruth[gt[0]] = gt[1]
    if dist is not None:
        dist.destroy_process_group()
    print("inference: %d videos in %.2f s on %d GPU(s)" % (len(data), time.time() - t0, world))
    path = os.path.join(output_dir, "VidORval_infer_results_{}.pkl".format(save_tag))
    with open(path, "wb") as f:
        pickle.dump(raw, f)
    print("infer_results have been saved at {}".format(path))
    return _report(groundtruth, predictions, output_dir, "VidORval_predict_relations_{}.json".format(save_tag), save_json, device)


def evaluate_combined(cfg_path, ckpt_path, cls_stage_result_path, save_tag="", output_dir=None, cuda=0, save_json=False, synthetic=0, n_trajs=32,
                      grd_batch=8):
    cfg = parse_config_py(cfg_path)
    ic = cfg["inference_config"]
    output_dir = output_dir or os.path.dirname(os.path.abspath(cfg_path))
    os.makedirs(output_dir, exist_ok=True)
    rank, world, device, dist = init_process(cuda)
    torch.manual_seed(0)
    model = DEBUG(ensure_grounding_files(cfg["model_config"], output_dir), is_train=False).to(device)
    if ckpt_path:
        load_checkpoint_into(model, ckpt_path, device)
    model.eval()
    data = _load_data(cfg, synthetic, n_trajs, cfg["model_config"]["dim_feat"], device)
    with open(cls_stage_result_path, "rb") as f:
        cls_results = pickle.load(f)
    predictions, groundtruth, n_skipped = {}, {}, 0
    t0 = time.time()
    th = (ic["score_th"], ic["tiou_th"], ic["bins_th"], ic["nms_th"])

    def flush(batch):
        """several videos per launch sequence (grd_forward_batch); the reference scores them one by one (tools/eval_vidor.py)"""
        nonlocal n_skipped
        if not batch:
            return
        jobs = [(clip.to(device), res[0].to(device), res[2].to(device), p.video_len) for p, res, clip in batch]
        with torch.no_grad():
            outs = model.forward_raw_batch(jobs, *th)
        for (p, res, clip), o in zip(batch, outs):
            if int(o["status"].item()) != 0:
                # the reference stops here (min() of an empty selection, models/grd_model_v5.py:726)
                print("skipping {}: temporal pooling selected no clip for some (query, bin): all scores are zero".format(p.video_name))
                n_skipped += 1
                continue
            out = (o["pooled_se"].cpu(), o["bins_probs"].cpu(), o["bins_mask"].bool().cpu())
            my_relations.update(E.to_eval_format(p.to(torch.device("cpu")), combine_with_grounding(tuple(t.cpu() for t in res), out, p.video_len),
                                                 dataset_type="vidor"))

    # the videos that have classification results, shared out by the work of the grounding stage (queries x clips); the share is
    # computed from the index (names + costs) and the classification results, and only this rank's videos are loaded
    names = data.names if data.names is not None else [data[i][0].video_name for i in range(len(data))]
    have = [i for i, nm in enumerate(names) if cls_results.get(nm) is not None and cls_results[nm][0] is not None and cls_results[nm][0].shape[0] > 0]
    costs = [int(cls_results[names[i]][0].shape[0]) * int(data.costs[i]) for i in have]
    from bigc_hip.shard import balanced_assignment
    mine = balanced_assignment(costs, world)[rank] if have else []
    rest = balanced_assignment([1] * len(names), world)[rank]          # ground truth of the synthetic videos: converted by an even split of ALL videos
    my_gt = {}
    if synthetic:
        for i in rest:
            p, g, _ = data[i]
            if g is not None:
                my_gt[E.reset_video_name(p.video_name, "vidor")] = E.gt_to_eval_format(g, dataset_type="vidor")
    my_todo = []
    for j in mine:
        p, g, clip = data[have[j]]
        my_todo.append((p, cls_results[p.video_name], clip))
    my_relations = {}
    for s0 in range(0, len(my_todo), grd_batch):
        flush(my_todo[s0:s0 + grd_batch])
    torch.cuda.synchronize()
    parts = gather_results([(my_relations, n_skipped, my_gt)], [rank], world, dist)       # one (relations, skipped, ground truth) triple per rank
    if parts is None:
        dist.destroy_process_group()
        return None
    n_skipped = 0
    for rel, sk, gt in parts:
        predictions.update(rel)
        groundtruth.update(gt)
        n_skipped += sk
    if dist is not None:
        dist.destroy_process_group()
    print("grounding: %d videos (%d skipped) in %.2f s on %d GPU(s)" % (len(data), n_skipped, time.time() - t0, world))
    return _report(groundtruth, predictions, output_dir, "VidORval_predict_relations_aft_grd_{}.json".format(save_tag), save_json, device)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="VidOR eval (classification stage / classification + grounding) on the MI355X HIP path")
    ap.add_argument("--eval_cls_only", action="store_true")
    ap.add_argument("--cfg_path", type=str)
    ap.add_argument("--ckpt_path", type=str, default=None)
    ap.add_argument("--output_dir", type=str, default=None)
    ap.add_argument("--cuda", type=int, default=0)
    ap.add_argument("--save_tag", type=str, default="")
    ap.add_argument("--save_json_results", action="store_true")
    ap.add_argument("--use_baseline", action="store_true")
    ap.add_argument("--json_results_path", type=str, default=None)
    ap.add_argument("--cls_stage_result_path", type=str, default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="evaluate on N generated videos instead of a dataset")
    ap.add_argument("--n_trajs", type=int, default=32)
    ap.add_argument("--grd_batch", type=int, default=8, help="videos per grounding launch sequence (1 = the reference's one video per forward)")
    ap.add_argument("--tracking_dir", type=str, default=None, help="directory of raw tracking results (<video>.npy): ingested by this package, no reference dataloader needed")
    ap.add_argument("--classeme_dir", type=str, default=None, help="with --tracking_dir: directory of the per-box classeme files (<video>_clsme.npy)")
    ap.add_argument("--video_feature_dir", type=str, default=None, help="with --tracking_dir: directory of the clip features of the grounding stage (<video>.npy)")
    ap.add_argument("--video_info_path", type=str, default=None, help="with --tracking_dir: JSON {video: {video_len, video_wh}} (default: last tracked frame + 1, 1280 x 720)")
    ap.add_argument("--gt_json_path", type=str, default=None, help="ground truth in the evaluator's JSON format (default: datasets/GT_json_for_eval/VidORval_gts.json)")
    a = ap.parse_args()
    assert not a.tracking_dir or a.classeme_dir, "--tracking_dir needs --classeme_dir"
    RAW.update(tracking_dir=a.tracking_dir, classeme_dir=a.classeme_dir, video_feature_dir=a.video_feature_dir, video_info_path=a.video_info_path,
               gt_json_path=a.gt_json_path)
    if a.json_results_path is not None:
        with open(a.json_results_path) as f:
            _report({}, json.load(f), ".", "", False)
    elif a.eval_cls_only:
        evaluate_cls_stage(a.cfg_path, a.ckpt_path, a.save_tag, a.output_dir, a.cuda, a.save_json_results, a.synthetic, a.n_trajs,
                           model_class=Base_C if a.use_baseline else BIG_C_vidor)        # tools/eval_vidor.py:310
    else:
        evaluate_combined(a.cfg_path, a.ckpt_path, a.cls_stage_result_path, a.save_tag, a.output_dir, a.cuda, a.save_json_results,
                          a.synthetic, a.n_trajs, a.grd_batch)
