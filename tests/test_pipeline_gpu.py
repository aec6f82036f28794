"""BASELINE config 3 (classification + grounding) end to end on the GPU: BIG-C (VidOR variant, 81/51 categories) ->
grounding scorer -> bin expansion -> relations, HIP models vs the CPU oracles through the same glue; Recall@50 parity."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cases
from bigc_hip import evaluate as E
from bigc_hip.pipeline import classify_then_ground, combine_with_grounding
from bigc_hip.synthetic import bigc_state_shapes, grd_state_shapes, seeded_state_dict, synthetic_gt_graph
from oracle import bigc_oracle as O
from oracle import grd_oracle as G

DEV = "cuda:0"


def test_classification_then_grounding_parity():
    """The classification stage is TRAINED first (HIP training step, learnable synthetic relation set of bigc_hip/recall_set.py), so
    that the Recall@50 compared between the HIP pipeline and the oracle pipeline is far from zero; the grounding scorer keeps
    seeded weights (it rescales scores and cuts spans; the relations it is given are the trained classifier's)."""
    import copy
    from models import BIG_C_vidor, DEBUG
    from bigc_hip import recall_set as R
    cfg = dict(R.CONFIG_SMALL_O)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidor"), 71, att_gain=1.0)
    gcfg = dict(dim_feat=32, dim_clsme=300, dim_hidden=64, num_bins=4, loss_factor={})
    gsd = seeded_state_dict(grd_state_shapes(gcfg), 72, 4.0)
    d = tempfile.mkdtemp()
    np.save(d + "/b.npy", sd["bias_matrix"].numpy())
    np.save(d + "/ge.npy", gsd["EntiNameEmb"].numpy())
    np.save(d + "/gp.npy", gsd["PredNameEmb"].numpy())
    torch.manual_seed(6)
    trainee = BIG_C_vidor(dict(cfg, bias_matrix_path=d + "/b.npy"), is_train=True)
    trainee.load_state_dict(sd)
    trainee = trainee.to(DEV)
    data = R.relational_set(cfg, 40000, 96, n_trajs=16, n_actors=3)
    for p, _ in data:
        p.to(DEV)
    R.train(trainee, data, 250, batch=16, lr=3e-4)
    sd = {k: v.detach().cpu() for k, v in trainee.state_dict().items()}
    bigc = BIG_C_vidor(dict(cfg, bias_matrix_path=d + "/b.npy"), is_train=False)
    bigc.load_state_dict(sd)
    bigc = bigc.to(DEV)
    grd = DEBUG(dict(gcfg, EntiNameEmb_path=d + "/ge.npy", PredNameEmb_path=d + "/gp.npy"), is_train=False)
    grd.load_state_dict(gsd)
    grd = grd.to(DEV)
    th = dict(score_th=0.9, tiou_th=0.5, bins_th=0.2, nms_th=0.8)
    hip, ora, cls_only = {}, {}, {}
    rs = np.random.RandomState(5)
    held = R.relational_set(cfg, 50000, 8, n_trajs=16, n_actors=3)
    gts = R.ground_truth(held)
    for p, _ in held:
        T = (p.video_len + 7) // 8
        clip = torch.from_numpy((0.1 * rs.standard_normal((T, gcfg["dim_feat"]))).astype(np.float32))
        # oracle pipeline
        r = O.forward_test(sd, cfg, [p], topk=3, variant="vidor")[0]
        cls_only.update(E.to_eval_format(p, r))
        o3 = None
        if r is not None:
            g = G.forward(gsd, gcfg, clip, (r[0], r[2], p.video_len), **th)
            o3 = combine_with_grounding(r, g, p.video_len)
        ora.update(E.to_eval_format(p, o3))
        # HIP pipeline
        h3 = classify_then_ground(bigc, grd, copy.deepcopy(p).to(DEV), clip.to(DEV), topk=3, **th)
        hip.update(E.to_eval_format(p, h3))
        if o3 is not None:
            assert h3 is not None
            assert torch.equal(h3[0], o3[0]) and torch.equal(h3[2], o3[2])
            assert float((h3[1] - o3[1]).abs().max()) < 5e-5
    m_h, r_h, p_h = E.evaluate(gts, hip)
    m_o, r_o, p_o = E.evaluate(gts, ora)
    _, r_c, _ = E.evaluate(gts, cls_only)
    print("Recall@50: classification only %.3f, with grounding oracle %.3f / HIP %.3f" % (r_c[50], r_o[50], r_h[50]))
    assert r_c[50] > 0.2, "the trained classifier should find relations: R@50 = %.3f" % r_c[50]
    assert r_o[50] > 0.2, "... and keep them through the grounding stage: R@50 = %.3f" % r_o[50]
    assert r_h[50] == r_o[50] and r_h[100] == r_o[100] and abs(m_h - m_o) < 1e-6
    assert sum(len(x) for x in hip.values()) > 0
