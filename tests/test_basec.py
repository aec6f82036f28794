"""The explicit pairwise baseline Base_C: CPU oracle and HIP path against outputs of the
reference's own ``models/model_pairwise_baseline.py`` (tests/golden/basec_*.npz)."""
import os
import tempfile

import numpy as np
import pytest
import torch

import cases
from oracle import basec_oracle as B


def _check(res, g, i, tol):
    if bool(g["v%d_none" % i]):
        assert res is None
        return
    assert res is not None
    assert np.array_equal(res[0].cpu().numpy(), g["v%d_quint" % i])                 # bit-exact indices, incl. the top-k cut order
    assert np.array_equal(res[2].cpu().numpy(), g["v%d_dura" % i])
    want = g["v%d_scores" % i]
    got = res[1].cpu().numpy()
    assert np.array_equal(got[:, 1:], want[:, 1:])                                  # tracklet scores are copies
    assert np.abs(got[:, 0] - want[:, 0]).max() < tol


@pytest.mark.parametrize("name", list(cases.BASEC_CASES))
def test_basec_oracle_matches_reference(name):
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    cfg, sd, props, topk = cases.make_basec_case(name)
    for i, p in enumerate(props):
        if p.num_proposals > 1:
            logits, _ = B.forward_propagation(sd, cfg, p)
            ref = g["v%d_logits" % i]
            assert np.abs(logits.numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    for i, r in enumerate(B.forward_test(sd, cfg, props, topk)):
        _check(r, g, i, 1e-6)


def _build(cfg, sd):
    from models import Base_C
    d = tempfile.mkdtemp()
    mc = dict(cfg)
    np.save(os.path.join(d, "b.npy"), sd["bias_matrix"].numpy())
    mc["bias_matrix_path"] = os.path.join(d, "b.npy")
    if "EntiNameEmb" in sd:
        np.save(os.path.join(d, "e.npy"), sd["EntiNameEmb"].numpy())
        mc["EntiNameEmb_path"] = os.path.join(d, "e.npy")
    m = Base_C(mc, is_train=False)
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.BASEC_CASES))
def test_basec_hip_matches_reference(name):
    from bigc_hip.packed import PackedBatch
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    cfg, sd, props, topk = cases.make_basec_case(name)
    model = _build(cfg, sd)
    dev_props = [p.to("cuda:0") for p in cases.make_basec_case(name)[2]]
    res = model(dev_props, topk=topk)
    assert len(res) == len(props)
    for i, r in enumerate(res):
        _check(r, g, i, 5e-5)
    # pair logits: 1e-4 relative (BASELINE north_star tolerance)
    keep = [p for p in dev_props if p.num_proposals > 0]
    out = model.engine().pair_forward(PackedBatch(keep, "cuda:0"))
    off, j = out["vid_pair_off"], 0
    for i, p in enumerate(props):
        if p.num_proposals == 0:
            continue
        if p.num_proposals > 1:
            ref = g["v%d_logits" % i]
            got = out["pair_logits"][off[j]:off[j + 1]].cpu().numpy()
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
        j += 1
