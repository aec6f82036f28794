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


def synthetic_source(model_config, n_videos, n_trajs, seed, with_gt=True, n_gt_trajs=8, n_preds=10):
    """Synthetic proposals (+ GT graphs) shaped by ``model_config`` (SURVEY.md section 8d recipe)."""
    from bigc_hip.synthetic import synthetic_gt_graph, synthetic_proposal
    width = model_config["dim_feat"] + (model_config.get("dim_i3d", None) or 0)
    out = []
    for v in range(n_videos):
        p = synthetic_proposal(seed + v, n_trajs=n_trajs, feat_width=width, num_enti_cats=model_config["num_enti_cats"],
                               name="synthetic_%05d" % v)
        g = synthetic_gt_graph(p, seed + 100000 + v, n_gt_trajs=n_gt_trajs, n_preds=n_preds,
                               num_pred_cats=model_config["num_pred_cats"]) if with_gt else None
        out.append((p, g))
    return out


def ensure_table_files(model_config, out_dir, seed=0):
    """The model constructors load ``bias_matrix_path`` / ``EntiNameEmb_path`` ``.npy`` files (models/model_0v10.py:264-286).
    For synthetic runs without the prepared data, seeded stand-ins are written next to the outputs."""
    cfg = dict(model_config)
    rs = np.random.RandomState(seed)
    ne, npred = cfg["num_enti_cats"], cfg["num_pred_cats"]
    if not (cfg.get("bias_matrix_path") and os.path.exists(cfg["bias_matrix_path"])):
        path = os.path.join(out_dir, "synthetic_bias_matrix.npy")
        np.save(path, rs.uniform(-7, 0, size=(ne, ne, npred)).astype(np.float32))
        cfg["bias_matrix_path"] = path
    if "EntiNameEmb_path" in cfg and cfg["EntiNameEmb_path"] and not os.path.exists(cfg["EntiNameEmb_path"]):
        path = os.path.join(out_dir, "synthetic_EntiNameEmb.npy")
        np.save(path, rs.normal(0, 0.4, size=(ne, cfg["dim_clsme"])).astype(np.float32))
        cfg["EntiNameEmb_path"] = path
    return cfg
