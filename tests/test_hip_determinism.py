"""Run-to-run bit reproducibility of the three training steps: EVERY parameter gradient and every loss term, not
only the 2-D weights.  The reductions that used float atomics until round 5 -- the bias gradients of the weight-gradient GEMMs
(csrc/backward.hip dw_bias_out), LayerNorm's dgamma / dbeta (partial_rows_finish_kernel), the bias_matrix scatter
(bias_matrix_bwd_kernel), the depth-wise conv parameter gradients of the grounding scorer (csrc/grounding_train.hip) and Base_C's BCE loss
(csrc/loss.hip bce_finish_kernel) -- now add per-workgroup partials in a fixed order.  Five runs in a row, larger dims where the
reductions really are split over many workgroups."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import cases
from test_basec import _build as build_basec
from test_hip_grd_train import build_model as build_grd
from test_hip_train import build_train_model

DEV = "cuda:0"
RUNS = 5


def _same(runs):
    first = runs[0]
    bad = []
    for i, r in enumerate(runs[1:], 1):
        assert set(r) == set(first)
        bad += [(i, k) for k in first if not torch.equal(first[k], r[k])]
    return bad


@pytest.mark.parametrize("name,train_mode", [("bigc_train_v", False), ("bigc_train_v64", False), ("bigo_train_o", False), ("bigc_train_v", True)])
def test_bigc_training_step_is_bit_reproducible(name, train_mode):
    """BIG-C (VidVRD and VidOR variants): loss terms + every gradient, dropout off and on (counter-based masks, same seed)"""
    if name not in cases.TRAIN_CASES:
        pytest.skip("no such case")
    runs = []
    for _ in range(RUNS):
        cfg, variant, sd, props, gts = cases.make_train_case(name)
        model = build_train_model(cfg, variant, sd)
        if train_mode:
            model.train()
        torch.manual_seed(77)
        total, terms = model([p.to(DEV) for p in props], gts)
        total.sum().backward()
        torch.cuda.synchronize()
        out = {"grad:" + k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        assert len(out) > 20 and any(k.endswith(".bias") for k in out) and "grad:bias_matrix" in out
        out["loss:total"] = total.detach().clone()
        out.update({"loss:" + k: v.detach().clone() for k, v in terms.items()})
        runs.append(out)
    assert not _same(runs), _same(runs)[:10]


@pytest.mark.parametrize("name", list(cases.BASEC_TRAIN_CASES))
def test_basec_training_step_is_bit_reproducible(name):
    runs = []
    for _ in range(RUNS):
        cfg, sd, props, pos, labels = cases.make_basec_train_case(name)
        model = build_basec(cfg, sd, is_train=True)
        total, terms = model([p.to(DEV) for p in props], [x.to(DEV) for x in pos], [x.to(DEV) for x in labels])
        total.backward()
        torch.cuda.synchronize()
        out = {"grad:" + k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        out["loss:total"] = total.detach().clone()
        runs.append(out)
    assert not _same(runs), _same(runs)[:10]


@pytest.mark.parametrize("name", list(cases.GRD_TRAIN_CASES))
def test_grounding_training_step_is_bit_reproducible(name):
    runs = []
    for _ in range(RUNS):
        cfg, sd, feats, gts, seed = cases.make_grd_train_case(name)
        model = build_grd(cfg, sd)
        torch.manual_seed(seed)
        total, terms = model([f.to(DEV) for f in feats], gts)
        total.backward()
        torch.cuda.synchronize()
        out = {"grad:" + k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        assert any("depth_wise" in k for k in out)
        out["loss:total"] = total.detach().clone()
        out.update({"loss:" + k: v.detach().clone() for k, v in terms.items()})
        runs.append(out)
    assert not _same(runs), _same(runs)[:10]


_DW_STREAM_SCRIPT = r"""
import os, sys
import numpy as np, torch
repo = os.environ["BIGC_REPO"]
sys.path.insert(0, os.path.join(repo, "tests")); sys.path.insert(0, os.path.join(repo, "tests", "golden"))
import cases
from test_hip_train import build_train_model
out = {}
for name in ("bigc_train_v64", "bigo_train_o"):
    cfg, variant, sd, props, gts = cases.make_train_case(name)
    model = build_train_model(cfg, variant, sd)
    total, terms = model([p.to("cuda:0") for p in props], gts)
    total.sum().backward()
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        if p.grad is not None:
            out[name + ":" + k] = p.grad.cpu().numpy()
    out[name + ":loss"] = total.detach().cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_decoder_weight_gradients_on_their_own_stream_give_the_same_bits():
    """Round 6: at big batches the decoder layers' grouped weight-gradient launches run on a stream of the handle's beside the next layer's
    backward chain (two buffer sets, fork / join by events; csrc/train.hip).  Forced on and forced off: every gradient, bit for bit."""
    import os, subprocess, sys, tempfile
    import numpy as np
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tempfile.mkdtemp()
    res = {}
    for tag in ("0", "1"):
        path = os.path.join(d, "dw_stream_" + tag + ".npz")
        r = subprocess.run([sys.executable, "-c", _DW_STREAM_SCRIPT, path], env=dict(os.environ, BIGC_REPO=repo, BIGC_DW_STREAM=tag),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = dict(np.load(path))
    assert set(res["0"]) == set(res["1"]) and len(res["0"]) > 60
    for k, v in res["0"].items():
        assert np.array_equal(v, res["1"][k], equal_nan=True), k
