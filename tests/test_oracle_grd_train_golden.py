"""Pin the grounding-training oracle (oracle/grd_train_oracle.py) against the reference's loss terms and gradients
(tests/golden/grd_train_*.npz); the unseeded negative sampling is reproduced by seeding the global torch RNG."""
import numpy as np
import pytest
import torch

import cases
from oracle import grd_train_oracle as GT


@pytest.mark.parametrize("name", list(cases.GRD_TRAIN_CASES))
def test_grd_train_oracle_matches_reference(name):
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    cfg, sd, feats, gts, seed = cases.make_grd_train_case(name)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.manual_seed(seed)
    total, terms, _ = GT.forward_train(sdg, cfg, feats, gts)
    total.backward()
    assert abs(float(total.detach()) - float(g["total"])) < 1e-5 * abs(float(g["total"]))
    for k, v in terms.items():
        assert abs(float(v.detach()) - float(g["loss_" + k])) < 1e-5 * max(1.0, abs(float(g["loss_" + k])))
    for k, v in sdg.items():
        want = float(g["gnorm_" + k])
        got = 0.0 if v.grad is None else float(v.grad.norm())
        assert abs(got - want) <= 5e-5 * max(want, 1e-6), k
        if "grad_" + k in g.files:
            ref = g["grad_" + k]
            assert np.abs(v.grad.numpy() - ref).max() <= 5e-5 * max(np.abs(ref).max(), 1e-12), k
