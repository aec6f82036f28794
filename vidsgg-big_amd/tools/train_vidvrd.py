#!/usr/bin/env python3
"""VidVRD BIG-C training on the HIP path; flags of the reference's ``tools/train_vidvrd.py`` (``--cfg_path
--output_dir --save_tag --use_pku --from_checkpoint --ckpt_path``).  One process per GPU: launch with
``python -m torch.distributed.run --nproc-per-node N tools/train_vidvrd.py ...`` for data parallelism (each rank
takes its contiguous chunk of every batch, gradients meet in ONE flat all-reduce over RCCL); a plain ``python``
launch trains on one GPU.  Loop body = tools/train_vidvrd.py:151-164; checkpoints carry the reference's keys
(``batch_size, crt_epoch, model_state_dict, optim_state_dict, sched_state_dict``, :20-38) with the ``module.`` prefix."""
import argparse
import os
import random

import torch

from _common import ensure_table_files, load_checkpoint_into, parse_config_py, synthetic_source

from bigc_hip.shard import allreduce_gradients, shard_indices
from models import BIG_C_vidvrd


def train(cfg_path, output_dir=None, save_tag="", use_pku=False, from_checkpoint=False, ckpt_path=None, synthetic=0, n_trajs=64, epochs=None):
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    cfg = parse_config_py(cfg_path)
    model_config, train_config = cfg["model_config"], cfg["train_config"]
    output_dir = output_dir or os.path.dirname(os.path.abspath(cfg_path))
    os.makedirs(output_dir, exist_ok=True)
    model_config = ensure_table_files(model_config, output_dir)
    torch.manual_seed(0)                                   # identical initial weights on every rank
    model = BIG_C_vidvrd(model_config, is_train=True).to(device)
    batch_size, total_epoch = train_config["batch_size"], epochs or train_config["total_epoch"]
    if synthetic:
        data = synthetic_source(model_config, synthetic, n_trajs, seed=4321)
    else:
        from dataloaders.dataloader_vidvrd import Dataset_pku_i3d, Dataset_pku, Dataset
        ds_cfg = cfg["train_dataset_config"]
        ds = (Dataset_pku_i3d if (use_pku and model_config.get("dim_i3d")) else Dataset_pku if use_pku else Dataset)(**ds_cfg)
        data = [ds[i] for i in range(len(ds))]
    n_iter = len(data) // batch_size                       # drop_last=True (:109)
    milestones = [int(m * len(data) / batch_size) for m in train_config["epoch_lr_milestones"]]
    optimizer = torch.optim.Adam(model.parameters(), lr=train_config["initial_lr"])
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma=train_config["lr_decay"])
    crt_epoch = 0
    if from_checkpoint:
        ckpt = load_checkpoint_into(model, ckpt_path, device)
        optimizer.load_state_dict(ckpt["optim_state_dict"])
        scheduler.load_state_dict(ckpt["sched_state_dict"])
        crt_epoch = ckpt["crt_epoch"]
    gt_duras = {id(g): g.traj_durations.clone() for _, g in data}
    it = 0
    for epoch in range(total_epoch):
        if epoch < crt_epoch:
            it += n_iter
            continue
        order = list(range(len(data)))
        random.Random(epoch).shuffle(order)                 # same permutation on every rank
        for b in range(n_iter):
            batch = sorted((data[i] for i in order[b * batch_size:(b + 1) * batch_size]), key=lambda pg: pg[1].max_frames)  # collator order (:469)
            mine = [batch[i] for i in shard_indices(len(batch), rank, world)]
            props = [p.to(device) for p, _ in mine]
            gts = [g for _, g in mine]
            for g in gts:
                g.traj_durations = gt_duras[id(g)].clone()  # keep the GT intact across epochs (the step decrements it in place)
            optimizer.zero_grad()
            loss, terms = model(props, gts)
            loss = loss.mean()
            loss.backward()
            allreduce_gradients(model.parameters(), dist)
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=5, norm_type=2)
            optimizer.step()
            scheduler.step()
            if rank == 0 and it % 10 == 0:
                print("epoch={},iter={},loss={:.4f}; ".format(epoch, it, float(loss)) +
                      "".join("{}:{:.4f}; ".format(k, float(v)) for k, v in terms.items()) + "lr={}".format(optimizer.param_groups[0]["lr"]))
            it += 1
        if rank == 0 and ((epoch + 1) % 10 == 0 or epoch + 1 == total_epoch):
            torch.save({"batch_size": batch_size, "crt_epoch": epoch + 1,
                        "model_state_dict": {"module." + k: v for k, v in model.state_dict().items()},
                        "optim_state_dict": optimizer.state_dict(), "sched_state_dict": scheduler.state_dict()},
                       os.path.join(output_dir, "model_epoch_{}_{}.pth".format(epoch + 1, save_tag)))
    if dist is not None:
        dist.destroy_process_group()
    return float(loss)


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
