"""The metrics code (bigc_hip/evaluate.py) against numbers produced by the reference's own evaluator
(tests/golden/evaluator.npz), and Recall@50 parity of the HIP path with the CPU oracle on synthetic GT."""
import os
import tempfile

import numpy as np
import pytest
import torch

import cases
from bigc_hip import evaluate as E
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_gt_graph


def test_metrics_match_reference_evaluator():
    g = np.load(os.path.join(cases.GOLDEN_DIR, "evaluator.npz"))
    for seed in cases.EVAL_SEEDS:
        gt, pr = cases.make_eval_case(seed)
        mean_ap, rec, prec = E.evaluate(gt, pr)
        got = np.array([mean_ap, rec[50], rec[100], prec[1], prec[5], prec[10]])
        assert np.allclose(got, g["s%d" % seed], rtol=0, atol=1e-7), (seed, got, g["s%d" % seed])
        g0, p0 = gt["video_0"][0], pr["video_0"][0]
        v = [E.viou(p0["sub_traj"], p0["duration"], g0["sub_traj"], g0["duration"]),
             E.viou(p0["obj_traj"], p0["duration"], g0["obj_traj"], g0["duration"])]
        assert np.allclose(v, g["s%d_viou" % seed], atol=1e-12)
    assert abs(E.voc_ap(np.array([0.1, 0.4, 0.4, 0.9]), np.array([1.0, 0.5, 0.66, 0.4])) - float(g["voc_ap"][0])) < 1e-12


def _synthetic_eval_set(cfg, n_videos, n_trajs):
    props, gts = [], {}
    for v in range(n_videos):
        p = cases.make_proposal(cfg, "vidvrd", dict(seed=3000 + v, n_trajs=n_trajs, len_range=(8, 40), start_range=(0, 20)))
        p.video_name = "synth_%d" % v
        g = synthetic_gt_graph(p, 4000 + v, n_gt_trajs=6, n_preds=8, num_pred_cats=cfg["num_pred_cats"], jitter=0.0)
        props.append(p)
        gts[p.video_name] = E.gt_to_eval_format(g)
    return props, gts


def test_result_conversion_and_oracle_metrics_cpu():
    """to_eval_format on the oracle's outputs: durations / cut trajectories are consistent and the metrics run"""
    from oracle import bigc_oracle as O
    cfg = cases.get_cfg("CONFIG_TINY")
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 61, att_gain=6.0)
    props, gts = _synthetic_eval_set(cfg, 4, 10)
    pred = {}
    for p, r in zip(props, O.forward_test(sd, cfg, props, topk=3)):
        rel = E.to_eval_format(p, r)
        for x in rel[p.video_name]:
            assert len(x["sub_traj"]) == len(x["obj_traj"]) == x["duration"][1] - x["duration"][0] > 0
        pred.update(rel)
    mean_ap, rec, prec = E.evaluate(gts, pred)
    assert 0.0 <= rec[50] <= rec[100] <= 1.0 and 0.0 <= mean_ap <= 1.0


def _trained_small_model(steps, n_train=96, n_held=12, n_actors=3, lr=3e-4, label_noise=0.0):
    """a BIG-C model of the exp2 structure at small dims, trained on the HIP training step on the learnable synthetic relation set
    (bigc_hip/recall_set.py) -> (eval model on the GPU, its state_dict on the host, config, held-out [(proposal, graph)] on the host)"""
    from models import BIG_C_vidvrd
    from bigc_hip import recall_set as R
    cfg = dict(R.CONFIG_SMALL)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 61, att_gain=1.0)
    d = tempfile.mkdtemp()
    np.save(d + "/b.npy", sd["bias_matrix"].numpy())
    np.save(d + "/e.npy", sd["EntiNameEmb"].numpy())
    mcfg = dict(cfg, bias_matrix_path=d + "/b.npy", EntiNameEmb_path=d + "/e.npy")
    torch.manual_seed(5)                                        # the dropout seeds of the training steps
    model = BIG_C_vidvrd(mcfg, is_train=True)
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    data = R.relational_set(cfg, 20000, n_train, n_trajs=16, n_actors=n_actors, label_noise=label_noise)
    for p, _ in data:
        p.to("cuda:0")
    R.train(model, data, steps, batch=16, lr=lr)
    ev = BIG_C_vidvrd(mcfg, is_train=False)
    ev.load_state_dict(model.state_dict())
    held = R.relational_set(cfg, 30000, n_held, n_trajs=16, n_actors=n_actors, label_noise=label_noise)
    return ev.to("cuda:0"), {k: v.detach().cpu() for k, v in model.state_dict().items()}, cfg, held


@pytest.mark.gpu
@pytest.mark.parametrize("steps,label_noise", [(60, 0.0), (250, 0.0), (250, 0.3)])
def test_recall_at_50_parity_hip_vs_oracle(steps, label_noise):
    """BASELINE metric, second half: Recall@50 of the HIP path equals the oracle's (|R@50(build) -
    R@50(oracle)| = 0 in fp32 parity mode) -- on a model that FINDS relations: trained for `steps` iterations on the HIP training
    step on the learnable synthetic relation set and scored on held-out videos, so that the recall compared is far from zero
    (with random weights and random predicates it is 0 on both sides and the statement is empty).  60 steps: a half-trained
    model, many relations near the top-50 cut; 250 steps: a trained one."""
    from bigc_hip import recall_set as R
    from oracle import bigc_oracle as O
    import copy
    ev, sd, cfg, held = _trained_small_model(steps, label_noise=label_noise, n_held=24 if label_noise else 12)
    gts = R.ground_truth(held)
    props = [p for p, _ in held]
    ora = R.relations(props, O.forward_test(sd, cfg, props, topk=5))
    hip = R.relations(props, ev([copy.deepcopy(p).to("cuda:0") for p in props], topk=5))
    m_h, r_h, p_h = E.evaluate(gts, hip)
    m_o, r_o, p_o = E.evaluate(gts, ora)
    assert r_o[50] > 0.2, "the trained model should find relations: R@50 = %.3f after %d steps" % (r_o[50], steps)
    if label_noise:
        # 30 % of the ground-truth predicates are random: the recall CANNOT saturate -- the gate compares numbers strictly inside (0, 1)
        assert r_o[50] < 0.95, "with label noise the recall should stay below 1: R@50 = %.3f" % r_o[50]
    assert r_h[50] == r_o[50] and r_h[100] == r_o[100]
    assert abs(m_h - m_o) < 1e-6 and p_h == p_o
    assert sum(len(v) for v in hip.values()) == sum(len(v) for v in ora.values()) > 0


def test_result_conversion_matches_reference_converter():
    """to_eval_format against the stored output of the reference's EvalFmtCvtor.to_eval_format_pr (tests/golden/evalfmt.json, made
    by make_golden.py): every relation dict -- dataset vocabulary (standard / PKU VidVRD numbering, VidOR), VidOR video-name
    reset, [start, end) durations, trajectories cut to the duration, mean score, background predicates dropped, None result."""
    import json
    want = json.load(open(os.path.join(cases.GOLDEN_DIR, "evalfmt.json")))
    for name, (dataset, use_pku, proposal, result) in cases.make_evalfmt_cases().items():
        got = E.to_eval_format(proposal, result, dataset_type=dataset, use_pku=use_pku)
        assert list(got) == list(want[name]), name
        for vid in got:
            g, w = got[vid], want[name][vid]
            assert len(g) == len(w), name
            for a, b in zip(g, w):
                assert a["triplet"] == b["triplet"] and list(a["duration"]) == list(b["duration"]), (name, a["triplet"], b["triplet"])
                assert abs(a["score"] - b["score"]) < 1e-7
                assert a["sub_traj"] == b["sub_traj"] and a["obj_traj"] == b["obj_traj"]
    nv, npred = E.name_tables("vidvrd", True)
    assert len(nv) == 36 and len(npred) == 133 and nv != E.name_tables("vidvrd", False)[0] and sorted(nv) == sorted(E.name_tables("vidvrd", False)[0])
    assert len(E.name_tables("vidor")[0]) == 81 and len(E.name_tables("vidor")[1]) == 51


@pytest.mark.gpu
def test_device_viou_table_matches_reference_viou():
    """the batched HIP vIoU kernel (bigc_viou_pairs) against the evaluator's per-pair host computation (common.py:65-106) on the
    evaluator fixtures, and the metrics computed through it against the reference evaluator's numbers"""
    g = np.load(os.path.join(cases.GOLDEN_DIR, "evaluator.npz"))
    n = 0
    for seed in cases.EVAL_SEEDS:
        gt, pr = cases.make_eval_case(seed)
        tables = E.viou_tables_device(gt, pr, "cuda:0")
        for vid, tab in tables.items():
            for (i, k), v in tab.items():
                p, q = pr[vid][i], gt[vid][k]
                assert tuple(p["triplet"]) == tuple(q["triplet"])
                ref = min(E.viou(p["sub_traj"], p["duration"], q["sub_traj"], q["duration"]),
                          E.viou(p["obj_traj"], p["duration"], q["obj_traj"], q["duration"]))
                assert abs(v - ref) < 2e-6, (vid, i, k, v, ref)
                n += 1
        mean_ap, rec, prec = E.evaluate(gt, pr, device="cuda:0")
        got = np.array([mean_ap, rec[50], rec[100], prec[1], prec[5], prec[10]])
        assert np.allclose(got, g["s%d" % seed], rtol=0, atol=1e-7), (seed, got, g["s%d" % seed])
    assert n > 20
