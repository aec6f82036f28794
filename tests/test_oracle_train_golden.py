"""Pin the training-step oracle (oracle/bigc_train_oracle.py) against the real reference's loss terms,
Hungarian indices and gradients stored in tests/golden/*train*.npz."""
import numpy as np
import pytest
import torch

import cases
from oracle import bigc_train_oracle as T


@pytest.mark.parametrize("name", list(cases.TRAIN_CASES))
def test_train_oracle_matches_reference(name):
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    cfg, variant, sd, props, gts = cases.make_train_case(name)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    before = [gt.traj_durations.clone() for gt in gts]
    total, terms, idx = T.forward_train(sdg, cfg, props, gts, variant)
    total.backward()
    assert abs(float(total) - float(g["total"])) < 1e-5 * abs(float(g["total"]))
    for k, v in terms.items():
        assert abs(float(v) - float(g["loss_" + k])) < 1e-5 * max(1.0, abs(float(g["loss_" + k])))
    for i, (iq, ig) in enumerate(idx):
        assert np.array_equal(iq.numpy(), g["match%d_query" % i]) and np.array_equal(ig.numpy(), g["match%d_gt" % i])
    for k, v in sdg.items():
        frozen = k == "EntiNameEmb" or (variant == "vidor" and k == "pos_embedding")   # requires_grad=False in the reference
        if "gnorm_" + k not in g.files or frozen:
            continue
        want = float(g["gnorm_" + k])
        got = 0.0 if v.grad is None else float(v.grad.norm())
        assert abs(got - want) <= 2e-5 * max(want, 1e-6), k
        if "grad_" + k in g.files:
            ref = g["grad_" + k]
            assert np.abs(v.grad.numpy() - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-12), k
    # the reference's in-place `-= 1` on the GT durations (models/model_0v10.py:567) is reproduced
    for gt, b in zip(gts, before):
        assert torch.equal(gt.traj_durations[:, 1], b[:, 1] - 1)
