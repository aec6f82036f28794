#!/usr/bin/env python3
"""VidVRD inference + evaluation on the HIP path; same flags as the reference's ``tools/eval_vidvrd.py:170-180``
(``--cfg_path --ckpt_path --output_dir --cuda --save_tag --use_pku --json_results_path``).

Data: with ``--synthetic N`` the videos and their ground truth are generated (no dataset is needed); otherwise
the reference's own ``dataloaders.dataloader_vidvrd`` must be importable (put the reference checkout AFTER this
package on PYTHONPATH) and is used exactly as the reference does (tools/eval_vidvrd.py:91-108).

Several GPUs: ``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 eval_vidvrd.py ...`` -- one process per
GPU, each scoring its size-balanced share of the videos (no data-path collective: videos are independent);
the per-video results are gathered on rank 0, which converts, saves and evaluates them exactly as the one-process run does
(replaces the reference's DataParallel wrapper, utils/DataParallel.py:45-113, around tools/eval_vidvrd.py:117-121)."""
import argparse
import json
import os
import pickle
import sys
import time

import torch

from _common import (VideoIndex, dataset_index, ensure_table_files, init_process, load_checkpoint_into, parse_config_py, shard_for_inference,
                     synthetic_index)

from bigc_hip import evaluate as E
from bigc_hip.shard import gather_results
from models import BIG_C_vidvrd


def inference_then_eval(cfg_path, ckpt_path=None, output_dir=None, cuda=0, save_tag="", use_pku=False, json_results_path=None,
                        synthetic=0, n_trajs=64, batch_videos=64, tracking_dir=None, i3d_dir=None, gt_json_path=None, video_info_path=None, cache_path=None):
    cfg = parse_config_py(cfg_path)
    model_config, infer_config = cfg["model_config"], cfg.get("inference_config", {})
    topk = infer_config.get("topk", 10)
    output_dir = output_dir or os.path.dirname(os.path.abspath(cfg_path))
    os.makedirs(output_dir, exist_ok=True)
    rank, world, device, dist = init_process(cuda)
    model_config = ensure_table_files(model_config, output_dir)
    torch.manual_seed(0)            # without --ckpt_path (synthetic smoke runs) every rank must draw the same initial weights
    model = BIG_C_vidvrd(model_config, is_train=False).to(device)
    if ckpt_path:
        load_checkpoint_into(model, ckpt_path, device)
    model.eval()
    # The data set as an INDEX (cost + loader per video): the assignment of videos to ranks is computed from it and every rank loads only
    # its own share (the reference iterates a DataLoader, tools/eval_vidvrd.py:91-121; an 8-GPU node reads the data set once, not 8 times)
    if synthetic:
        data = synthetic_index(model_config, synthetic, n_trajs, seed=1234)
    elif cache_path:
        # the reference's own on-disk cache (datasets/cache/*.pkl, dataloaders/dataloader_vidvrd.py:170-183: pickled instances of ITS classes),
        # read by a restricted unpickler into this package's containers -- no reference code needed
        from bigc_hip.ingest import load_reference_cache
        cached = load_reference_cache(cache_path)
        names = sorted(cached)
        data = VideoIndex([max(1, sum(cached[n][0]._lens) if cached[n][0].num_proposals else 1) for n in names], lambda i: cached[names[i]][:2], names)
    elif tracking_dir:
        # raw tracking results (one `.npy` of per-box rows per video, PKU format): ingested here, feature rows assembled on the GPU
        import glob
        import numpy as np
        from bigc_hip.ingest import load_pku_tracklets
        ds_cfg = cfg["test_dataset_config"]
        paths = sorted(glob.glob(os.path.join(tracking_dir, "*.npy")))
        names = [os.path.splitext(os.path.basename(path))[0] for path in paths]
        info = json.load(open(video_info_path)) if video_info_path else {}      # {video: {video_len, video_wh}}; default: last tracked frame + 1, 1280 x 720

        def load(i):
            i3d = np.load(os.path.join(i3d_dir, names[i] + ".npy")) if i3d_dir else None
            vi = info.get(names[i], {})
            return (load_pku_tracklets(names[i], np.load(paths[i], allow_pickle=True), ds_cfg["dim_boxfeature"], ds_cfg["min_frames_th"],
                                       ds_cfg["max_proposal"], i3d_res=i3d, dim_i3d=ds_cfg.get("dim_i3d", 0) if i3d_dir else 0,
                                       device=device, video_len=vi.get("video_len"), video_wh=tuple(vi.get("video_wh", (1280, 720)))), None)
        data = VideoIndex([max(1, os.path.getsize(path)) for path in paths], load, names)       # cost = file size ~ rows
    else:
        from dataloaders.dataloader_vidvrd import Dataset_pku_i3d, Dataset_pku, Dataset   # the reference's own loaders
        ds_cfg = cfg["test_dataset_config"]
        ds = (Dataset_pku_i3d if (use_pku and model_config.get("dim_i3d")) else Dataset_pku if use_pku else Dataset)(**ds_cfg)
        data = dataset_index(ds)
    predictions, groundtruth, raw = {}, {}, {}
    # Real data: predictions are named with the dataset's vocabulary -- the PKU proposals number the categories differently
    # (utils/evaluate.py:79-82) -- and evaluated against datasets/GT_json_for_eval/VidVRDtest_gts.json as eval_relation_with_gt
    # does.  Synthetic videos have no JSON: their ground truth comes from the generated graphs, which share the proposals'
    # numbering, so both sides use the standard table.
    pku_names = use_pku and not synthetic
    cache_gt = bool(cache_path) and not gt_json_path          # a cache carries the videos' ground-truth graphs: used unless a JSON is given
    # this rank's share of the videos, balanced by the number of feature rows (sum of the tracklet lengths), loaded now
    mine, my_data = shard_for_inference(data, data.costs, rank, world)
    t0 = time.time()
    my_results = []
    # two chunks in flight: the host side of chunk i + 1 (moving it to the GPU, the batch tables, the launches) runs while the GPU works on
    # chunk i (BigcBase.forward_async); the results are those of model(props, topk)
    chunks = [my_data[s:s + batch_videos] for s in range(0, len(my_data), batch_videos)]
    pending = None
    for ci in range(len(chunks) + 1):
        upcoming = None
        if ci < len(chunks):
            props = [p.to(device) for p, _ in chunks[ci]]
            with torch.no_grad():
                upcoming = model.forward_async(props, topk=topk)      # one packed HIP forward for the whole chunk, queued
        if pending is None:
            pending = upcoming
            continue
        chunk, results = chunks[ci - 1], pending.result()
        pending = upcoming
        for (p, g), r in zip(chunk, results):
            # converted to the evaluator's format HERE, by the rank that holds the video's tracklets: rank 0 receives per video
            # (raw result, prediction entry, ground-truth entry) and never needs the other ranks' proposals
            r = None if r is None else tuple(t.cpu() for t in r)
            pred = E.to_eval_format(p.to(torch.device("cpu")), r, dataset_type="vidvrd", use_pku=pku_names)
            gt = E.gt_to_eval_format(g, dataset_type="vidvrd") if (g is not None and (synthetic or cache_gt)) else None
            my_results.append((p.video_name, r, pred, gt))
    torch.cuda.synchronize()
    all_results = gather_results(my_results, mine, len(data), dist)
    if all_results is None:                                 # ranks other than 0 are done
        dist.destroy_process_group()
        return None
    for name, r, pred, gt in all_results:
        predictions.update(pred)
        raw[name] = None if r is None else list(r)
        if gt is not None:
            groundtruth[name] = gt
    infer_s = time.time() - t0
    with open(os.path.join(output_dir, "VidVRDtest_infer_result_%s.pkl" % save_tag), "wb") as f:
        pickle.dump(raw, f)
    if json_results_path:
        with open(json_results_path, "w") as f:
            json.dump(predictions, f)
    if not groundtruth:
        gt_json = gt_json_path or "datasets/GT_json_for_eval/VidVRDtest_gts.json"      # VidVRDhelperEvalAPIs/visual_relation_detection.py:250-253
        with open(gt_json) as f:
            groundtruth = json.load(f)
    mean_ap, rec, prec = E.evaluate(groundtruth, predictions, device=device)     # vIoU overlaps from one batched HIP kernel
    print("inference: %d videos in %.2f s on %d GPU(s) (this rank loaded %d of them)" % (len(data), infer_s, world, data.n_loaded))
    print("detection mean AP (used in challenge): {}".format(mean_ap))
    print("detection recall: {}".format(rec))
    print("tagging precision: {}".format(prec))
    if dist is not None:
        dist.destroy_process_group()
    return mean_ap, rec, prec


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="VidVRD eval on the MI355X HIP path")
    ap.add_argument("--cfg_path", type=str, required=True)
    ap.add_argument("--ckpt_path", type=str, default=None)
    ap.add_argument("--output_dir", type=str, default=None)
    ap.add_argument("--cuda", type=int, default=0)
    ap.add_argument("--save_tag", type=str, default="")
    ap.add_argument("--use_pku", action="store_true")
    ap.add_argument("--json_results_path", type=str, default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="evaluate on N generated videos instead of a dataset")
    ap.add_argument("--n_trajs", type=int, default=64)
    ap.add_argument("--tracking_dir", type=str, default=None, help="directory of raw per-video tracking-result .npy files (PKU row format)")
    ap.add_argument("--i3d_dir", type=str, default=None, help="matching directory of per-box I3D feature .npy files")
    ap.add_argument("--cache_path", type=str, default=None, help="a datasets/cache/*.pkl file written by the reference's Dataset (read without the reference on PYTHONPATH)")
    ap.add_argument("--video_info_path", type=str, default=None, help="with --tracking_dir: JSON {video: {video_len, video_wh}} (default: last tracked frame + 1, 1280 x 720)")
    ap.add_argument("--gt_json_path", type=str, default=None, help="ground truth in the evaluator's JSON format (default: datasets/GT_json_for_eval/VidVRDtest_gts.json)")
    a = ap.parse_args()
    inference_then_eval(a.cfg_path, a.ckpt_path, a.output_dir, a.cuda, a.save_tag, a.use_pku, a.json_results_path, a.synthetic, a.n_trajs,
                        tracking_dir=a.tracking_dir, i3d_dir=a.i3d_dir, gt_json_path=a.gt_json_path, video_info_path=a.video_info_path, cache_path=a.cache_path)
