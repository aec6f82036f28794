"""Shared pieces of the tools/ entry points (config parsing, checkpoints, data sources)."""
import os
import runpy
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
if PKG not in sys.path:
    sys.path.insert(0, PKG)


def parse_config_py(path):
    """Every top-level name of a config ``.py`` as a dict (the reference imports a temp copy of the file,
    utils/utils_func.py:15-42; ``runpy`` gives the same mapping)."""
    assert path.endswith(".py")
    return {k: v for k, v in runpy.run_path(os.path.abspath(os.path.expanduser(path))).items() if not k.startswith("__")}


def strip_module_prefix(state_dict):
    """Checkpoints saved from the DataParallel wrapper carry a 7-char ``module.`` prefix
    (tools/eval_vidvrd.py:81-87, conditional form tools/eval_vidor.py:59-68)."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def load_checkpoint_into(model, ckpt_path, device):
    ckpt = torch.load(ckpt_path, map_location=device)
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    model.load_state_dict(strip_module_prefix(sd))
    return ckpt


class VideoIndex(object):
    """What a rank needs to take its share of a data set WITHOUT loading it: a cost and (where cheaply known) a name per video, and a
    loader.  The reference iterates a DataLoader (tools/eval_vidvrd.py:91-121, tools/train_vidvrd.py:96-110), so a process only ever
    holds the videos it works on; here the assignment of videos to ranks is computed from this index and ``index[i]`` loads video i on
    first use -- an 8-GPU node reads and holds the data set once, not eight times.  ``keep``: how many loaded videos to hold on to
    (None = all: evaluation touches each of its videos once; training passes a bound)."""

    def __init__(self, costs, load, names=None, keep=None):
        self.costs, self.names, self._load, self._keep = list(costs), names, load, keep
        self._cache = {}
        self.n_loaded = 0                                      # videos this process actually loaded (tests assert the share)

    def __len__(self):
        return len(self.costs)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __getitem__(self, i):
        if not 0 <= i < len(self.costs):
            raise IndexError(i)
        if i not in self._cache:
            if self._keep is not None and len(self._cache) >= self._keep:
                self._cache.pop(next(iter(self._cache)))       # oldest first
            self._cache[i] = self._load(i)
            self.n_loaded += 1
        return self._cache[i]


def _synthetic_lens_cost(seed, n_trajs, len_range=(8, 64)):
    """rows of ``synthetic_proposal(seed, ...)`` without generating it: its first draw is the tracklet lengths"""
    return int(np.random.RandomState(seed).randint(len_range[0], len_range[1], size=n_trajs).sum()) if n_trajs else 1


def synthetic_index(model_config, n_videos, n_trajs, seed, with_gt=True, n_gt_trajs=8, n_preds=10, keep=None):
    """Synthetic proposals (+ GT graphs) shaped by ``model_config`` (SURVEY.md section 8d recipe), generated on demand."""
    from bigc_hip.synthetic import synthetic_gt_graph, synthetic_proposal
    width = model_config["dim_feat"] + (model_config.get("dim_i3d", None) or 0)

    def load(v):
        p = synthetic_proposal(seed + v, n_trajs=n_trajs, feat_width=width, num_enti_cats=model_config["num_enti_cats"],
                               name="synthetic_%05d" % v)
        g = synthetic_gt_graph(p, seed + 100000 + v, n_gt_trajs=n_gt_trajs, n_preds=n_preds,
                               num_pred_cats=model_config["num_pred_cats"]) if with_gt else None
        return (p, g)

    return VideoIndex([_synthetic_lens_cost(seed + v, n_trajs) for v in range(n_videos)], load, ["synthetic_%05d" % v for v in range(n_videos)], keep)


def synthetic_source(model_config, n_videos, n_trajs, seed, with_gt=True, n_gt_trajs=8, n_preds=10):
    """the same videos, all of them, as a list"""
    idx = synthetic_index(model_config, n_videos, n_trajs, seed, with_gt, n_gt_trajs, n_preds)
    return [idx[v] for v in range(n_videos)]


def dataset_index(ds, convert=None, keep=None):
    """One of the reference's ``torch.utils.data.Dataset`` classes (dataloaders/dataloader_vidvrd.py, dataloader_vidor_v3.py): ``ds[i]`` is
    the loader; costs are unknown before loading (1 each: the assignment is then an even split by count)."""
    return VideoIndex([1] * len(ds), (lambda i: convert(ds[i])) if convert else (lambda i: ds[i]), None, keep)


def _save_npy_atomic(path, array):
    """write-then-rename: several ranks of a torch.distributed.run job may produce the same (seeded, identical) stand-in file at the same
    time, and a reader must never see a half-written one (seen as a flaky `EOFError: No data left in file` in the two-rank tool tests)"""
    tmp = "%s.%d.tmp.npy" % (path, os.getpid())
    np.save(tmp, array)
    os.replace(tmp, path)


def ensure_table_files(model_config, out_dir, seed=0):
    """The model constructors load ``bias_matrix_path`` / ``EntiNameEmb_path`` ``.npy`` files (models/model_0v10.py:264-286).
    For synthetic runs without the prepared data, seeded stand-ins are written next to the outputs."""
    cfg = dict(model_config)
    rs = np.random.RandomState(seed)
    ne, npred = cfg["num_enti_cats"], cfg["num_pred_cats"]
    if not (cfg.get("bias_matrix_path") and os.path.exists(cfg["bias_matrix_path"])):
        path = os.path.join(out_dir, "synthetic_bias_matrix.npy")
        _save_npy_atomic(path, rs.uniform(-7, 0, size=(ne, ne, npred)).astype(np.float32))
        cfg["bias_matrix_path"] = path
    if "EntiNameEmb_path" in cfg and cfg["EntiNameEmb_path"] and not os.path.exists(cfg["EntiNameEmb_path"]):
        path = os.path.join(out_dir, "synthetic_EntiNameEmb.npy")
        _save_npy_atomic(path, rs.normal(0, 0.4, size=(ne, cfg["dim_clsme"])).astype(np.float32))
        cfg["EntiNameEmb_path"] = path
    return cfg


def synthetic_vidor_index(model_config, n_videos, n_trajs, seed, clip_dim=None, len_range=(15, 120), n_gt_trajs=8, n_preds=10, keep=None):
    """VidOR-shaped synthetic videos: RoI feature + 300-d classeme columns when ``use_clsme`` (models/model_0v7.py:331-340),
    and -- with ``clip_dim`` -- the clip features the grounding stage reads, ``T = ceil(video_len / 8)`` rows
    (16-frame clips with overlap 0.5, experiments/grounding_weights/config_.py:22).  Items (proposal, gt_graph, clip), on demand."""
    from bigc_hip.synthetic import synthetic_gt_graph, synthetic_proposal
    width = model_config.get("dim_feat", 1024) + (model_config.get("dim_clsme", 300) if model_config.get("use_clsme", True) else 0)

    def load(v):
        p = synthetic_proposal(seed + v, n_trajs=n_trajs, feat_width=width, num_enti_cats=model_config.get("num_enti_cats", 81),
                               len_range=len_range, start_range=(0, len_range[1]), name="synthetic_%05d" % v)
        g = synthetic_gt_graph(p, seed + 100000 + v, n_gt_trajs=n_gt_trajs, n_preds=n_preds, num_pred_cats=model_config.get("num_pred_cats", 51))
        clip = None
        if clip_dim:
            rs = np.random.RandomState(seed + 200000 + v)
            clip = torch.from_numpy((0.1 * rs.standard_normal((-(-p.video_len // 8), clip_dim))).astype(np.float32))
        return (p, g, clip)

    return VideoIndex([_synthetic_lens_cost(seed + v, n_trajs, len_range) for v in range(n_videos)], load, ["synthetic_%05d" % v for v in range(n_videos)], keep)


def synthetic_vidor_source(model_config, n_videos, n_trajs, seed, clip_dim=None, len_range=(15, 120), n_gt_trajs=8, n_preds=10):
    idx = synthetic_vidor_index(model_config, n_videos, n_trajs, seed, clip_dim, len_range, n_gt_trajs, n_preds)
    return [idx[v] for v in range(n_videos)]


def ensure_grounding_files(model_config, out_dir, seed=0):
    """``DEBUG`` loads ``EntiNameEmb_path`` (81,300) and ``PredNameEmb_path`` (51,300) (models/grd_model_v5.py:158-170)."""
    cfg = dict(model_config)
    rs = np.random.RandomState(seed)
    for key, rows in (("EntiNameEmb_path", 81), ("PredNameEmb_path", 51)):
        if not (cfg.get(key) and os.path.exists(cfg[key])):
            path = os.path.join(out_dir, "synthetic_" + key[:-5] + ".npy")
            _save_npy_atomic(path, rs.normal(0, 0.4, size=(rows, cfg["dim_clsme"])).astype(np.float32))
            cfg[key] = path
    return cfg


def init_process(cuda=None):
    """One process per GPU: (rank, world, device, dist-or-None) from the torch.distributed.run environment.  Rank r uses GPU
    LOCAL_RANK (``cuda`` when started without a launcher).  BIGC_DIST_BACKEND=gloo with BIGC_DIST_ONE_GPU=1 puts every rank on GPU 0
    and exchanges through the host: how the two-rank paths are tested on a one-GPU box (RCCL cannot put two ranks on one device)."""
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("BIGC_DIST_BACKEND", "nccl")
    if world == 1 and cuda is not None:
        local = cuda
    if os.environ.get("BIGC_DIST_ONE_GPU", "0") == "1":
        local = 0
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world, device, dist


def shard_for_inference(data, costs, rank, world):
    """This rank's videos of an inference run: size-balanced over the ranks (bigc_hip.shard.balanced_assignment).
    -> (indices into ``data``, the items).  ``data`` may be a ``VideoIndex``: only this rank's videos are then loaded."""
    from bigc_hip.shard import balanced_assignment
    mine = balanced_assignment(costs, world)[rank]
    return mine, [data[i] for i in mine]


def training_loop(model, data, train_config, output_dir, save_tag, step, rank, world, device, dist, sort_key=None,
                  from_checkpoint=False, ckpt_path=None, epochs=None):
    """The reference's epoch / iteration loop (tools/train_vidvrd.py:121-181, tools/train_vidor.py:590-633) for one rank:
    Adam + MultiStepLR on iteration milestones, ``drop_last`` batches, every rank takes its chunk of each batch
    (utils/DataParallel.py:80-113), ONE flat gradient all-reduce, ``clip_grad_norm_(5)``, checkpoints in the reference's
    format with the ``module.`` key prefix.  ``step(model, items)`` -> (loss, terms)."""
    import random
    from bigc_hip.shard import allreduce_gradients, clip_grad_norm_, shard_indices, zero_gradients
    batch_size, total_epoch = train_config["batch_size"], epochs or train_config["total_epoch"]
    n_iter = len(data) // batch_size
    milestones = [int(m * len(data) / batch_size) for m in train_config["epoch_lr_milestones"]]
    optimizer = torch.optim.Adam(model.parameters(), lr=train_config["initial_lr"], fused=device.type == "cuda")
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma=train_config["lr_decay"])
    crt_epoch = 0
    if from_checkpoint:
        ckpt = load_checkpoint_into(model, ckpt_path, device)
        optimizer.load_state_dict(ckpt["optim_state_dict"])
        scheduler.load_state_dict(ckpt["sched_state_dict"])
        crt_epoch = ckpt["crt_epoch"]
    it, loss = 0, torch.zeros(())
    for epoch in range(total_epoch):
        if epoch < crt_epoch:
            it += n_iter
            continue
        order = list(range(len(data)))
        random.Random(epoch).shuffle(order)                 # same permutation on every rank
        for b in range(n_iter):
            ids = order[b * batch_size:(b + 1) * batch_size]
            if isinstance(data, VideoIndex) and world > 1:
                # each rank loads ITS chunk of the batch only.  The collator orders a batch by the GT graphs' max_frames (sort_key,
                # dataloaders/dataloader_vidvrd.py:469), which would need every video of the batch loaded on every rank; the order
                # matters -- it decides which videos share a replica, and each replica's cls_pos / cls_neg / adj terms are ratios
                # over its OWN n_pos / n_neg / n_adj, so the mean of the per-replica terms depends on the grouping --
                # so the batch is ordered by the index's per-video cost instead (the row count: what the index knows without
                # loading a video; ties keep the shuffled order).  Equally valid, but NOT the reference's composition of the
                # replicas, nor the world == 1 path's on the same seed: multi-rank runs follow their own gradient sequence.
                ids = sorted(ids, key=lambda i: data.costs[i])
                batch = ids
                mine = [data[ids[i]] for i in shard_indices(len(ids), rank, world)]
            else:
                batch = [data[i] for i in ids]
                if sort_key is not None:
                    batch = sorted(batch, key=sort_key)
                mine = [batch[i] for i in shard_indices(len(batch), rank, world)]
            optimizer.zero_grad()
            if mine:
                loss, terms = step(model, mine)
                loss = loss.mean()
                loss.backward()
            else:                                           # batch smaller than the number of ranks: nothing to compute here, but the
                zero_gradients(list(model.parameters()))    # collective below needs every rank
                loss, terms = torch.zeros((), device=device), {}
            flat = allreduce_gradients(model.parameters(), dist, contributing=min(len(batch), world))
            clip_grad_norm_(model.parameters(), 5, flat=flat)
            optimizer.step()
            scheduler.step()
            if rank == 0 and it % 10 == 0:
                print("epoch={},iter={},loss={:.4f}; ".format(epoch, it, float(loss)) +
                      "".join("{}:{:.4f}; ".format(k, float(v)) for k, v in terms.items()) + "lr={}".format(optimizer.param_groups[0]["lr"]),
                      flush=True)
            it += 1
        if rank == 0 and ((epoch + 1) % 10 == 0 or epoch + 1 == total_epoch):
            torch.save({"batch_size": batch_size, "crt_epoch": epoch + 1,
                        "model_state_dict": {"module." + k: v for k, v in model.state_dict().items()},
                        "optim_state_dict": optimizer.state_dict(), "sched_state_dict": scheduler.state_dict()},
                       os.path.join(output_dir, "model_epoch_{}_{}.pth".format(epoch + 1, save_tag)))
    if dist is not None:
        dist.destroy_process_group()
    return float(loss)
