#!/usr/bin/env python3
"""VidVRD BIG-C training on the HIP path; flags of the reference's ``tools/train_vidvrd.py`` (``--cfg_path
--output_dir --save_tag --use_pku --from_checkpoint --ckpt_path``).  One process per GPU: launch with
``python -m torch.distributed.run --nproc-per-node N tools/train_vidvrd.py ...`` for data parallelism (each rank
takes its contiguous chunk of every batch, gradients meet in ONE flat all-reduce over RCCL); a plain ``python``
launch trains on one GPU.  Loop body = tools/train_vidvrd.py:151-164; checkpoints carry the reference's keys
(``batch_size, crt_epoch, model_state_dict, optim_state_dict, sched_state_dict``, :20-38) with the ``module.`` prefix."""
import argparse
import os

import torch

from _common import dataset_index, ensure_table_files, init_process, parse_config_py, synthetic_index, training_loop

from models import BIG_C_vidvrd


def train(cfg_path, output_dir=None, save_tag="", use_pku=False, from_checkpoint=False, ckpt_path=None, synthetic=0, n_trajs=64, epochs=None):
    rank, world, device, dist = init_process()
    cfg = parse_config_py(cfg_path)
    model_config, train_config = cfg["model_config"], cfg["train_config"]
    output_dir = output_dir or os.path.dirname(os.path.abspath(cfg_path))
    os.makedirs(output_dir, exist_ok=True)
    model_config = ensure_table_files(model_config, output_dir)
    torch.manual_seed(0)                                   # identical initial weights on every rank
    model = BIG_C_vidvrd(model_config, is_train=True).to(device)
    # the data set as an index: every rank loads the videos of ITS chunks only, when it first needs them (the reference's DataLoader
    # workers do the same, tools/train_vidvrd.py:96-110); at most one epoch's share is held
    if synthetic:
        data = synthetic_index(model_config, synthetic, n_trajs, seed=4321, keep=None if world == 1 else max(8, 2 * synthetic // world))
    else:
        from dataloaders.dataloader_vidvrd import Dataset_pku_i3d, Dataset_pku, Dataset
        ds_cfg = cfg["train_dataset_config"]
        ds = (Dataset_pku_i3d if (use_pku and model_config.get("dim_i3d")) else Dataset_pku if use_pku else Dataset)(**ds_cfg)
        data = dataset_index(ds, keep=None if world == 1 else max(8, 2 * len(ds) // world))

    def step(model, items):
        props = [p.to(device) for p, _ in items]
        gts = [g for _, g in items]
        for g in gts:
            if not hasattr(g, "_dura0"):
                g._dura0 = g.traj_durations.clone()         # as loaded
            g.traj_durations = g._dura0.clone()             # keep the GT intact across epochs (the step decrements it in place)
        return model(props, gts)

    return training_loop(model, data, train_config, output_dir, save_tag, step, rank, world, device, dist,
                         sort_key=lambda pg: pg[1].max_frames,          # collator order (dataloaders/dataloader_vidvrd.py:469)
                         from_checkpoint=from_checkpoint, ckpt_path=ckpt_path, epochs=epochs)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="VidVRD BIG-C training on the MI355X HIP path")
    ap.add_argument("--cfg_path", type=str, required=True)
    ap.add_argument("--output_dir", type=str, default=None)
    ap.add_argument("--save_tag", type=str, default="")
    ap.add_argument("--use_pku", action="store_true")
    ap.add_argument("--from_checkpoint", action="store_true")
    ap.add_argument("--ckpt_path", type=str, default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="train on N generated videos instead of a dataset")
    ap.add_argument("--n_trajs", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=None)
    a = ap.parse_args()
    train(a.cfg_path, a.output_dir, a.save_tag, a.use_pku, a.from_checkpoint, a.ckpt_path, a.synthetic, a.n_trajs, a.epochs)
