"""Training-mode dropout of the HIP path: the counter-based mask generator against its numpy mirror, the attention core
with dropout (forward + backward) against torch autograd with the same masks, and a whole BIG-C training step with dropout
against the CPU oracle applying identical masks."""
import math

import numpy as np
import pytest
import torch

import cases
from bigc_hip import _lib
from oracle import dropout_ref as DR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def S():
    return _lib.current_stream()


def test_mask_generator_matches_numpy_mirror():
    lib = _lib.load()
    rows, n, ld, p, seed, site = 37, 50, 52, 0.1, 123456789, 4113
    x = torch.randn(rows, ld)
    xd = x.to(DEV)
    _lib.check(lib.bigc_dropout_f32(S(), xd.data_ptr(), rows, n, ld, p, seed, site))
    idx = np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(n) + np.arange(n, dtype=np.uint64)[None, :]
    keep = torch.from_numpy(DR.keep_mask(seed, site, idx, p))
    want = x.clone()
    want[:, :n] = torch.where(keep, x[:, :n] * np.float32(DR.scale_of(p)), torch.zeros(()))
    assert torch.equal(xd.cpu(), want)                           # padding columns untouched, kept values scaled, rest zero
    frac = 1.0 - keep.float().mean().item()
    assert abs(frac - p) < 0.03


@pytest.mark.parametrize("H,hd,ql", [(2, 16, [5, 40]), (8, 64, [70, 33]), (4, 8, [3]), (8, 64, [64, 20]), (2, 64, [192, 1, 161])])
def test_attention_core_with_dropout_fwd_bwd(H, hd, ql):
    lib = _lib.load()
    g = torch.Generator().manual_seed(sum(ql) + hd)
    d, p, seed, site = H * hd, 0.25, 77, 4096 + 16
    off = np.concatenate([[0], np.cumsum(ql)]).astype(np.int32)
    q, k, v = (torch.randn(sum(ql), d, generator=g, requires_grad=True) for _ in range(3))
    go = torch.randn(sum(ql), d, generator=g)
    drop = DR.Dropper(p, seed)
    outs = []
    for s in range(len(ql)):
        a, b = off[s], off[s + 1]
        qs = q[a:b].reshape(-1, H, hd).transpose(0, 1)
        ks = k[a:b].reshape(-1, H, hd).transpose(0, 1)
        vs = v[a:b].reshape(-1, H, hd).transpose(0, 1)
        att = drop.att(site, torch.softmax(qs @ ks.transpose(1, 2) / math.sqrt(hd), -1), int(a))
        outs.append((att @ vs).transpose(0, 1).reshape(-1, d))
    want = torch.cat(outs)
    want.backward(go)
    qd, kd, vd, god = (t.detach().to(DEV) for t in (q, k, v, go))
    offd = torch.from_numpy(off).to(DEV)
    o = torch.empty(sum(ql), d, device=DEV)
    lse = torch.empty(sum(ql), H, device=DEV)
    _lib.check(lib.bigc_mha_core_drop_f32(S(), qd.data_ptr(), d, kd.data_ptr(), d, vd.data_ptr(), d, o.data_ptr(), d, offd.data_ptr(),
                                          offd.data_ptr(), len(ql), max(ql), H, hd, lse.data_ptr(), p, seed, site))
    dq, dk, dv = (torch.empty_like(qd) for _ in range(3))
    delta = torch.empty(sum(ql), H, device=DEV)
    _lib.check(lib.bigc_mha_bwd_drop_f32(S(), qd.data_ptr(), d, kd.data_ptr(), d, vd.data_ptr(), d, o.data_ptr(), d, god.data_ptr(), d,
                                         lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), d, dk.data_ptr(), d, dv.data_ptr(), d,
                                         offd.data_ptr(), offd.data_ptr(), len(ql), max(ql), max(ql), sum(ql), H, hd, p, seed, site))
    torch.cuda.synchronize()
    rel = lambda x, y: float((x.cpu() - y).abs().max() / y.abs().max())
    assert rel(o, want.detach()) < 1e-5
    assert rel(dq, q.grad) < 2e-5 and rel(dk, k.grad) < 2e-5 and rel(dv, v.grad) < 2e-5


@pytest.mark.parametrize("name", ["bigc_train_tiny", "bigo_train_tiny"])
def test_training_step_with_dropout_matches_oracle_with_same_masks(name):
    from oracle import bigc_train_oracle as T
    from test_hip_train import build_train_model
    cfg, variant, sd, props, gts = cases.make_train_case(name)
    model = build_train_model(cfg, variant, sd).train()           # dropout on
    model.dropout_p = 0.2
    torch.manual_seed(99)
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())       # the draw HipTrainStep.forward makes
    torch.manual_seed(99)
    total, terms = model([p.to(DEV) for p in cases.make_train_case(name)[3]], cases.make_train_case(name)[4])
    total.backward()
    sdg = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    o_total, o_terms, _ = T.forward_train(sdg, cfg, props, gts, variant, drop=DR.Dropper(0.2, seed))
    o_total.backward()
    assert abs(float(total) - float(o_total)) < 2e-4 * abs(float(o_total))
    # and it really is a different function from the dropout-free step
    e_total, _, _ = T.forward_train(sd, cfg, cases.make_train_case(name)[3], cases.make_train_case(name)[4], variant)
    assert abs(float(e_total) - float(o_total)) > 1e-3 * abs(float(o_total))
    for k, p in model.named_parameters():
        ref = sdg[k].grad
        if ref is None or not p.requires_grad:
            continue
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= 1e-3 * max(float(ref.abs().max()), 1e-12), (k, err, float(ref.abs().max()))
