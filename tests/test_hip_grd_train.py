"""Training step of the grounding scorer on the HIP path against the reference's
loss terms and gradients (tests/golden/grd_train_*.npz) and, parameter by parameter, against the CPU oracle."""
import os
import tempfile

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build_model(cfg, sd, is_train=True):
    from models import DEBUG
    d = tempfile.mkdtemp()
    cfg = dict(cfg)
    np.save(os.path.join(d, "e.npy"), sd["EntiNameEmb"].numpy())
    np.save(os.path.join(d, "p.npy"), sd["PredNameEmb"].numpy())
    cfg["EntiNameEmb_path"], cfg["PredNameEmb_path"] = os.path.join(d, "e.npy"), os.path.join(d, "p.npy")
    m = DEBUG(cfg, is_train=is_train)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()        # parity with the reference is defined with dropout off (is_train selects the training PATH)


def hip_step(name, weights=None):
    cfg, sd, feats, gts, seed = cases.make_grd_train_case(name)
    model = build_model(cfg, sd)
    torch.manual_seed(seed)
    total, terms = model([f.to(DEV) for f in feats], gts)
    if weights is None:
        total.backward()
    else:
        sum(w * terms[k] for k, w in weights.items()).backward()
    return model, total, terms


@pytest.mark.parametrize("name", list(cases.GRD_TRAIN_CASES))
def test_grd_train_matches_reference_golden(name):
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    model, total, terms = hip_step(name)
    assert abs(float(total) - float(g["total"])) < 1e-4 * abs(float(g["total"]))
    for k, v in terms.items():
        assert abs(float(v) - float(g["loss_" + k])) < 1e-4 * max(1.0, abs(float(g["loss_" + k]))), k
    for k, p in model.named_parameters():
        want = float(g["gnorm_" + k])
        got = 0.0 if p.grad is None else float(p.grad.norm())
        assert abs(got - want) <= 5e-4 * max(want, 1e-6), (k, got, want)
        if "grad_" + k in g.files:
            ref = g["grad_" + k]
            assert np.abs(p.grad.cpu().numpy() - ref).max() <= 5e-4 * max(np.abs(ref).max(), 1e-12), k


@pytest.mark.parametrize("weights", [None, dict(pos_cls=2.0, regr=0.5, neg_ct=1.5)])
def test_grd_train_every_gradient_matches_oracle(weights):
    from oracle import grd_train_oracle as GT
    name = "grd_train_tiny"
    cfg, sd, feats, gts, seed = cases.make_grd_train_case(name)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.manual_seed(seed)
    total, terms, _ = GT.forward_train(sdg, cfg, feats, gts)
    (total if weights is None else sum(w * terms[k] for k, w in weights.items())).backward()
    model, h_total, h_terms = hip_step(name, weights)
    for k in terms:
        assert abs(float(h_terms[k]) - float(terms[k])) < 1e-4 * max(1.0, abs(float(terms[k]))), k
    for k, p in model.named_parameters():
        ref = sdg[k].grad
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= 5e-4 * max(float(ref.abs().max()), 1e-12), (k, err, float(ref.abs().max()))


def test_grd_eval_with_gt_data_uses_the_gt_queries():
    from bigc_hip.grd_train_step import gt_queries
    cfg, sd, feats, gts, _ = cases.make_grd_train_case("grd_train_tiny")
    model = build_model(cfg, sd, is_train=False)
    feat = feats[0].to(DEV)
    pooled, probs, mask = model([feat], [gts[0]], with_gt_data=True)
    quint, dura, _, _ = gt_queries(gts[0], False)
    pooled2, probs2, mask2 = model([feat], [(quint, dura, gts[0].video_len)], with_gt_data=False)
    assert pooled.shape[0] == quint.shape[0]
    assert torch.equal(pooled, pooled2) and torch.equal(probs, probs2) and torch.equal(mask, mask2)


def test_grd_train_with_dropout_matches_oracle_with_same_masks():
    """training-mode dropout of the QANet layers: HIP step vs the oracle applying identical counter-based masks"""
    from oracle import dropout_ref as DR
    from oracle import grd_train_oracle as GT
    name = "grd_train_tiny"
    cfg, sd, feats, gts, seed = cases.make_grd_train_case(name)
    model = build_model(cfg, sd).train()
    torch.manual_seed(seed)
    total, terms = model([f.to(DEV) for f in feats], gts)
    total.backward()
    # replay the global RNG: per video, the negatives are drawn first, then the dropout seed
    import bigc_hip.grd_train_step as GS
    torch.manual_seed(seed)
    seeds = []
    for g in cases.make_grd_train_case(name)[3]:
        GS.gt_queries(g, True, 51)
        seeds.append(int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.manual_seed(seed)
    drops = [DR.Dropper(0.1, s) for s in seeds]
    # the oracle draws the negatives of video i itself; consume the dropout-seed draw after each video as the HIP host does
    orig = GT.prepare_gt_queries

    def patched(sd_, g, train=True, n_pred_cats=51):
        out = orig(sd_, g, train, n_pred_cats)
        torch.randint(0, 2 ** 31 - 1, (1,))          # consume what the HIP host drew for the dropout seed
        return out
    GT.prepare_gt_queries = patched
    try:
        o_total, o_terms, _ = GT.forward_train(sdg, cfg, feats, cases.make_grd_train_case(name)[3], drops=drops)
    finally:
        GT.prepare_gt_queries = orig
    o_total.backward()
    assert abs(float(total) - float(o_total)) < 2e-4 * abs(float(o_total))
    for k, p in model.named_parameters():
        ref = sdg[k].grad
        if ref is None:
            continue
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= 1e-3 * max(float(ref.abs().max()), 1e-12), (k, err, float(ref.abs().max()))


_STREAMS_SCRIPT = r"""
import os, sys
import numpy as np, torch
repo = os.environ["BIGC_REPO"]
sys.path.insert(0, os.path.join(repo, "tests")); sys.path.insert(0, os.path.join(repo, "tests", "golden"))
import cases
from test_hip_grd_train import build_model, DEV
from bigc_hip.synthetic import synthetic_gt_graph
from cases import synthetic_proposal
cfg, sd, _, _, seed = cases.make_grd_train_case("grd_train_g")
model = build_model(cfg, sd)
rs = np.random.RandomState(5)
feats, gts = [], []
for v, T in enumerate([40, 23, 61, 9, 33]):
    p = synthetic_proposal(3100 + v, n_trajs=10, feat_width=4, num_enti_cats=81, len_range=(30, 64), start_range=(0, 20))
    gts.append(synthetic_gt_graph(p, 3200 + v, n_gt_trajs=6, n_preds=7, num_pred_cats=51))
    feats.append(torch.from_numpy((0.1 * rs.standard_normal((T, cfg["dim_feat"]))).astype(np.float32)).to(DEV))
out = {}
for step in range(3):                      # the first step of a model runs on one stream (it records the weights a step prepares)
    model.zero_grad()
    torch.manual_seed(seed)
    total, terms = model(feats, gts)
    total.backward()
    torch.cuda.synchronize()
    if step:
        for k, p in model.named_parameters():
            if p.grad is not None:
                out["s%d:%s" % (step, k)] = p.grad.cpu().numpy()
        out["s%d:loss" % step] = np.asarray([float(total)] + [float(terms[k]) for k in sorted(terms)])
np.savez(sys.argv[1], **out)
"""


def test_videos_of_a_training_batch_on_their_own_streams():
    """Round 6: from a model's second step on, the videos of a batch run on streams of the engine's (forward: video 0 on the caller's
    stream, where the step's weights are prepared, the others forked behind it; backward: one flat gradient buffer per stream, added in a
    fixed order).  Two runs give the same bits; one stream (BIGC_GRD_TRAIN_STREAMS=1: the sums in another order) the same values to f32
    rounding; steps 2 and 3 of a run (same weights, same seed) the same bits."""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tempfile.mkdtemp()
    res = {}
    for tag, env in (("a", {}), ("b", {}), ("one", {"BIGC_GRD_TRAIN_STREAMS": "1"})):
        path = os.path.join(d, tag + ".npz")
        r = subprocess.run([sys.executable, "-c", _STREAMS_SCRIPT, path], env=dict(os.environ, BIGC_REPO=repo, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = dict(np.load(path))
    assert len(res["a"]) > 100
    for k, v in res["a"].items():
        assert np.array_equal(v, res["b"][k], equal_nan=True), k
        if k.startswith("s1:"):
            assert np.array_equal(v, res["a"]["s2:" + k[3:]], equal_nan=True), k
        w = res["one"][k]
        assert np.abs(v - w).max() <= 2e-5 * max(np.abs(w).max(), 1e-12), (k, float(np.abs(v - w).max()), float(np.abs(w).max()))
