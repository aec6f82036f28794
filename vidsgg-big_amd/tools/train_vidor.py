#!/usr/bin/env python3
"""VidOR training on the HIP path with the flags of the reference's ``tools/train_vidor.py:642-651``
(``--cfg_path --from_checkpoint --ckpt_path --train_baseline --train_grounding --output_dir --save_tag``):
BIG-C classification stage by default (``train_cls_stage`` :357-495), the temporal grounding scorer with
``--train_grounding`` (``train_grounding_stage`` :497-636), the pairwise baseline with ``--train_baseline`` (:175-352).  One process per GPU
(``python -m torch.distributed.run --nproc-per-node N tools/train_vidor.py ...``); see ``_common.training_loop``."""
import argparse
import os

import torch

from _common import (VideoIndex, dataset_index, ensure_grounding_files, ensure_table_files, init_process, parse_config_py, synthetic_vidor_index,
                     synthetic_vidor_source, training_loop)

from models import BIG_C_vidor, DEBUG


def _dataset(cfg, world=1, convert=None):
    """the reference's data set as an index: a rank loads the videos of its own chunks only (_common.VideoIndex)"""
    from dataloaders.dataloader_vidor import Dataset                  # the reference's loader (tools/train_vidor.py:15)
    ds = Dataset(**cfg["train_dataset_config"])
    return dataset_index(ds, convert, keep=None if world == 1 else max(8, 2 * len(ds) // world))


def _mapped(index, fn):
    """the same index with every loaded item passed through ``fn``"""
    return VideoIndex(index.costs, lambda i: fn(index[i]), index.names, index._keep)


def train_cls_stage(cfg_path, output_dir=None, save_tag="", from_checkpoint=False, ckpt_path=None, synthetic=0, n_trajs=32, epochs=None):
    rank, world, device, dist = init_process()
    cfg = parse_config_py(cfg_path)
    output_dir = output_dir or os.path.dirname(os.path.abspath(cfg_path))
    os.makedirs(output_dir, exist_ok=True)
    model_config = ensure_table_files(cfg["model_config"], output_dir)
    torch.manual_seed(0)
    model = BIG_C_vidor(model_config, is_train=True).to(device)
    keep = None if world == 1 else max(8, 2 * synthetic // world)
    data = _mapped(synthetic_vidor_index(model_config, synthetic, n_trajs, seed=8765, keep=keep), lambda it: (it[0], it[1])) if synthetic \
        else _dataset(cfg, world)

    def step(model, items):
        gts = [g for _, g in items]
        for g in gts:
            if not hasattr(g, "_dura0"):
                g._dura0 = g.traj_durations.clone()         # as loaded
            g.traj_durations = g._dura0.clone()             # the step decrements the GT end frames in place (models/model_0v7.py, as 0v10:567)
        return model([p.to(device) for p, _ in items], gts)

    return training_loop(model, data, cfg["train_config"], output_dir, save_tag, step, rank, world, device, dist,
                         sort_key=lambda pg: pg[1].max_frames, from_checkpoint=from_checkpoint, ckpt_path=ckpt_path, epochs=epochs)


def train_baseline(cfg_path, output_dir=None, save_tag="", from_checkpoint=False, ckpt_path=None, synthetic=0, n_trajs=24, epochs=None):
    """``train_baseline`` of the reference (:175-352): Base_C on the proposal pairs that hit a GT predicate
    (``prop_pair_to_gt_pred`` :78-170, vIoU threshold ``model_config["positive_vIoU_th"]``), BCE over multi-hot labels."""
    from bigc_hip import _lib
    from bigc_hip.packed import PackedBatch
    from bigc_hip.pair_train_step import assign_pair_labels
    from models import Base_C
    rank, world, device, dist = init_process()
    cfg = parse_config_py(cfg_path)
    output_dir = output_dir or os.path.dirname(os.path.abspath(cfg_path))
    os.makedirs(output_dir, exist_ok=True)
    model_config = ensure_table_files(cfg["model_config"], output_dir)
    torch.manual_seed(0)
    model = Base_C(model_config, is_train=True).to(device)
    pairs = [(p, g) for p, g, _ in synthetic_vidor_source(model_config, synthetic, n_trajs, seed=7654)] if synthetic else _dataset(cfg)
    lib = _lib.load()
    data, th = [], model_config.get("positive_vIoU_th", 0.5)
    for p, g in pairs:                                        # label assignment once, as the reference's cached label_map (:226-258)
        if p.num_proposals < 2 or g.num_trajs == 0 or g.num_preds == 0:
            continue
        ids, multihot = assign_pair_labels(lib, PackedBatch([p.to(device)], device), [g], th, model_config["num_pred_cats"])[0]
        if ids is not None:
            data.append((p, ids, multihot))
    if rank == 0:
        print("Base_C: %d of %d videos have labelled pairs (%d pairs)" % (len(data), len(pairs), sum(int(d[1].shape[0]) for d in data)), flush=True)

    def step(model, items):
        return model([p.to(device) for p, _, _ in items], [i for _, i, _ in items], [m for _, _, m in items])

    return training_loop(model, data, cfg["train_config"], output_dir, save_tag, step, rank, world, device, dist,
                         from_checkpoint=from_checkpoint, ckpt_path=ckpt_path, epochs=epochs)


def train_grounding_stage(cfg_path, output_dir=None, save_tag="", from_checkpoint=False, ckpt_path=None, synthetic=0, n_trajs=16, epochs=None):
    rank, world, device, dist = init_process()
    cfg = parse_config_py(cfg_path)
    output_dir = output_dir or os.path.dirname(os.path.abspath(cfg_path))
    os.makedirs(output_dir, exist_ok=True)
    model_config = ensure_grounding_files(cfg["model_config"], output_dir)
    torch.manual_seed(0)
    model = DEBUG(model_config, is_train=True).to(device)
    if synthetic:
        keep = None if world == 1 else max(8, 2 * synthetic // world)
        data = _mapped(synthetic_vidor_index({}, synthetic, n_trajs, seed=9876, clip_dim=model_config["dim_feat"], keep=keep),
                       lambda it: (it[2], it[0], it[1]))
    else:
        data = _dataset(cfg, world)                          # (video_feature, proposal, gt_graph) with video_feature_dir set
    torch.manual_seed(1000 + rank)                          # negative predicates are drawn from the global RNG (models/grd_model_v5.py:285-303)

    def step(model, items):
        return model([f.to(device) for f, _, _ in items], [g for _, _, g in items])

    return training_loop(model, data, cfg["train_config"], output_dir, save_tag, step, rank, world, device, dist,
                         from_checkpoint=from_checkpoint, ckpt_path=ckpt_path, epochs=epochs)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="VidOR training (BIG-C or grounding stage) on the MI355X HIP path")
    ap.add_argument("--cfg_path", type=str, required=True)
    ap.add_argument("--from_checkpoint", action="store_true")
    ap.add_argument("--ckpt_path", type=str, default=None)
    ap.add_argument("--train_baseline", action="store_true")
    ap.add_argument("--train_grounding", action="store_true")
    ap.add_argument("--output_dir", type=str, default=None)
    ap.add_argument("--save_tag", type=str, default="")
    ap.add_argument("--synthetic", type=int, default=0, help="train on N generated videos instead of a dataset")
    ap.add_argument("--n_trajs", type=int, default=None)
    ap.add_argument("--epochs", type=int, default=None)
    a = ap.parse_args()
    kw = dict(output_dir=a.output_dir, save_tag=a.save_tag, from_checkpoint=a.from_checkpoint, ckpt_path=a.ckpt_path, synthetic=a.synthetic,
              epochs=a.epochs)
    if a.n_trajs:
        kw["n_trajs"] = a.n_trajs
    (train_baseline if a.train_baseline else train_grounding_stage if a.train_grounding else train_cls_stage)(a.cfg_path, **kw)
