#!/usr/bin/env python3
"""Does the learnable synthetic relation set (bigc_hip/synthetic.py: relational_video) train?  exp2-dims BIG-C on the HIP training
step, Recall@50 / mAP of the HIP eval path on held-out videos every `--every` steps.
    python benchmarks/train_relational.py --steps 300 --batch 16 --lr 1e-4"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT):
    sys.path.insert(0, p)
from bigc_hip import configs, evaluate as E  # noqa: E402
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, relational_video  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--train-videos", type=int, default=64)
    ap.add_argument("--eval-videos", type=int, default=16)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--att-gain", type=float, default=1.0)
    ap.add_argument("--marker", type=float, default=3.0)
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--actors", type=int, default=4)
    ap.add_argument("--dropout", type=int, default=1)
    a = ap.parse_args()
    from models import BIG_C_vidvrd
    dev = "cuda:0"
    if a.tiny:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import cases
        cfg = cases.get_cfg("CONFIG_TINY")
    elif a.small:
        cfg = dict(configs.CONFIG_V, num_enti_cats=9, num_pred_cats=20, dim_ffn=64, dim_enti=64, dim_pred=64, dim_att=64, dim_feat=128, dim_clsme=12,
                   dim_i3d=32, n_enco_layers=2, n_deco_layers=3, n_att_head=4, num_querys=32)
    else:
        cfg = dict(configs.CONFIG_V)
    width = cfg["dim_feat"] + (cfg.get("dim_i3d") or 0)
    mw = 128 if not (a.tiny or a.small) else max(4, cfg["dim_feat"] // 8)
    n_trajs = 64 if not (a.tiny or a.small) else 16
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 11, att_gain=a.att_gain)
    d = tempfile.mkdtemp()
    np.save(d + "/b.npy", sd["bias_matrix"].numpy()); np.save(d + "/e.npy", sd["EntiNameEmb"].numpy())
    mcfg = dict(cfg, bias_matrix_path=d + "/b.npy", EntiNameEmb_path=d + "/e.npy")
    model = BIG_C_vidvrd(mcfg, is_train=True)
    model.load_state_dict(sd)
    model = model.to(dev)
    if not a.dropout:
        model.eval()
    ev = BIG_C_vidvrd(mcfg, is_train=False).to(dev)
    mk = lambda s: relational_video(s, n_trajs=n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"], num_pred_cats=cfg["num_pred_cats"],
                                    marker=a.marker, marker_width=mw, n_actors=a.actors)
    t0 = time.time()
    train = [mk(20000 + v) for v in range(a.train_videos)]
    held = [mk(30000 + v) for v in range(a.eval_videos)]
    gts_eval = {p.video_name: E.gt_to_eval_format(g) for p, g in held}
    for p, _ in train + held:
        p.to(dev)
    duras = {id(g): g.traj_durations.clone() for _, g in train}
    print("data %.1f s" % (time.time() - t0), flush=True)
    opt = torch.optim.Adam(model.parameters(), lr=a.lr, fused=True)
    rs = np.random.RandomState(0)

    gts_train = None

    def evaluate(on_train=False):
        nonlocal gts_train
        ev.load_state_dict(model.state_dict())
        preds = {}
        data = train[:a.eval_videos] if on_train else held
        if on_train and gts_train is None:
            for _, g in data:
                g.traj_durations = duras[id(g)].clone()
            gts_train = {p.video_name: E.gt_to_eval_format(g) for p, g in data}
        n_rel = 0
        for (p, _), r in zip(data, ev([p for p, _ in data], topk=10)):
            preds.update(E.to_eval_format(p, r))
            n_rel += 0 if r is None else int(r[0].shape[0])
        m, r, pr = E.evaluate(gts_train if on_train else gts_eval, preds)
        return m, r[50], r[100], n_rel

    t0 = time.time()
    for it in range(a.steps + 1):
        if it % a.every == 0:
            torch.cuda.synchronize()
            m, r50, r100, n_rel = evaluate()
            mt, rt50, rt100, nt = evaluate(True)
            print("step %4d  %.1f s  held-out mAP %.4f R@50 %.4f R@100 %.4f (%d relations) | train videos mAP %.4f R@50 %.4f (%d relations)" % (it, time.time() - t0, m, r50, r100, n_rel, mt, rt50, nt), flush=True)
        if it == a.steps:
            break
        pick = rs.choice(len(train), size=a.batch, replace=False)
        props, gts = [train[i][0] for i in pick], [train[i][1] for i in pick]
        for g in gts:
            g.traj_durations = duras[id(g)].clone()
        opt.zero_grad()
        loss, terms = model(props, gts)
        loss.mean().backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
        opt.step()
        if it % a.every == 0:
            print("      loss %.4f  " % float(loss) + "  ".join("%s %.4f" % (k, float(v)) for k, v in terms.items()), flush=True)


if __name__ == "__main__":
    main()
