"""GPU parity of the BIG-C training step on the HIP path: every backward kernel against torch autograd on
the CPU, then loss terms / Hungarian indices / gradients of the whole step against the real reference's
stored values (tests/golden/*train*.npz)."""
import math
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import cases
from bigc_hip import _lib
from oracle import bigc_oracle as O
from oracle import bigc_train_oracle as T

DEV = "cuda:0"


def S():
    return _lib.current_stream()


def dev(t):
    return t.to(DEV).contiguous()


def i32(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.int32).to(DEV)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


@pytest.mark.parametrize("M,N,K", [(1000, 16, 8), (5000, 512, 512), (777, 133, 3160), (300, 40, 96)])
def test_linear_dw(M, N, K):
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N)
    dY = torch.randn(M, N, generator=g)
    X = torch.randn(M, K, generator=g)
    dW0 = torch.randn(N, K, generator=g)
    db0 = torch.randn(N, generator=g)
    dW, db = dev(dW0), dev(db0)
    a, b = dev(dY), dev(X)
    _lib.check(lib.bigc_linear_dw_f32(S(), a.data_ptr(), N, b.data_ptr(), K, None, 1, K, dW.data_ptr(), K, db.data_ptr(), M, N, K))
    torch.cuda.synchronize()
    assert rel(dW, dW0 + dY.t() @ X) < 1e-5
    assert rel(db, db0 + dY.sum(0)) < 1e-5


def test_linear_dw_gather():
    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    src = torch.randn(60, 32, generator=g)
    M, nseg, N = 500, 3, 24
    idx = torch.randint(-1, 60, (M, nseg), generator=g)
    dY = torch.randn(M, N, generator=g)
    A = torch.zeros(M, nseg, 32)
    for s in range(nseg):
        ok = idx[:, s] >= 0
        A[ok, s] = src[idx[ok, s]]
    want = dY.t() @ A.reshape(M, -1)
    dW = torch.zeros(N, nseg * 32, device=DEV)
    a, b, ii = dev(dY), dev(src), i32(idx.numpy())
    _lib.check(lib.bigc_linear_dw_f32(S(), a.data_ptr(), N, b.data_ptr(), 32, ii.data_ptr(), nseg, 32, dW.data_ptr(), nseg * 32, None, M, N, nseg * 32))
    torch.cuda.synchronize()
    assert rel(dW, want) < 1e-5


@pytest.mark.parametrize("rows,d", [(5, 16), (3000, 512), (17, 128)])
def test_layernorm_bwd(rows, d):
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, d, generator=g) * 2 + 0.5).requires_grad_(True)
    w = torch.randn(d, generator=g).requires_grad_(True)
    b = torch.randn(d, generator=g).requires_grad_(True)
    dy = torch.randn(rows, d, generator=g)
    F.layer_norm(x, (d,), w, b, 1e-5).backward(dy)
    dx = torch.empty(rows, d, device=DEV)
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    xd, wd, dyd = dev(x.detach()), dev(w.detach()), dev(dy)
    _lib.check(lib.bigc_layernorm_bwd_f32(S(), xd.data_ptr(), wd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, d, 1e-5))
    torch.cuda.synchronize()
    assert rel(dx, x.grad) < 2e-5 and rel(dg, w.grad) < 2e-5 and rel(db, b.grad) < 2e-5


@pytest.mark.parametrize("hd,H,lens", [(64, 8, [64, 192]), (8, 2, [6, 1, 40]), (16, 8, [33, 70])])
def test_mha_bwd(hd, H, lens):
    lib = _lib.load()
    g = torch.Generator().manual_seed(hd + len(lens))
    d, n = hd * H, sum(lens)
    qkv = torch.randn(n, 3 * d, generator=g).requires_grad_(True)
    dO = torch.randn(n, d, generator=g)
    off = np.concatenate([[0], np.cumsum(lens)])
    outs = []
    for s in range(len(lens)):
        a, b = off[s], off[s + 1]
        q = qkv[a:b, :d].reshape(-1, H, hd).transpose(0, 1)
        k = qkv[a:b, d:2 * d].reshape(-1, H, hd).transpose(0, 1)
        v = qkv[a:b, 2 * d:].reshape(-1, H, hd).transpose(0, 1)
        outs.append((torch.softmax(q @ k.transpose(1, 2) / math.sqrt(hd), -1) @ v).transpose(0, 1).reshape(-1, d))
    out = torch.cat(outs)
    out.backward(dO)
    qd, dod, offd = dev(qkv.detach()), dev(dO), i32(off)
    o = torch.empty(n, d, device=DEV)
    lse = torch.empty(n, H, device=DEV)
    _lib.check(lib.bigc_mha_core_f32(S(), qd.data_ptr(), 3 * d, qd.data_ptr() + 4 * d, 3 * d, qd.data_ptr() + 8 * d, 3 * d, o.data_ptr(), d,
                                     offd.data_ptr(), offd.data_ptr(), len(lens), max(lens), H, hd, lse.data_ptr()))
    dqkv = torch.full((n, 3 * d), float("nan"), device=DEV)
    delta = torch.empty(n, H, device=DEV)
    _lib.check(lib.bigc_mha_bwd_f32(S(), qd.data_ptr(), 3 * d, qd.data_ptr() + 4 * d, 3 * d, qd.data_ptr() + 8 * d, 3 * d, o.data_ptr(), d,
                                    dod.data_ptr(), d, lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), 3 * d, dqkv.data_ptr() + 4 * d, 3 * d,
                                    dqkv.data_ptr() + 8 * d, 3 * d, offd.data_ptr(), offd.data_ptr(), len(lens), max(lens), max(lens), n, H, hd))
    torch.cuda.synchronize()
    assert rel(o, out) < 5e-6
    assert not torch.isnan(dqkv).any()
    assert rel(dqkv, qkv.grad) < 2e-5


@pytest.mark.parametrize("gmag,vmag", [(1e-9, 1.0), (1.0, 1.0), (3e5, 1.0), (1.0, 1e5)])
def test_mha_bwd_sharp_softmax_and_tiny_or_huge_upstream_gradient(gmag, vmag):
    """head_dim 64 runs on the split-operand f16 MFMAs (csrc/attention_bwd_split.hip), exact only above an absolute floor, while
    d scores = p (dP - delta) cancels under a sharp softmax: the kernels bring dO to a fixed magnitude first (1e-9: the encoder's
    upstream gradient late in training; without it the second training step's conv gradients were 19 % off; 3e5: an unscaled seed), and
    an operand out of the f16 range (values of 1e5) raises the launch's flag: the f32 kernels behind redo it.  Truth: autograd in f64."""
    lib = _lib.load()
    hd, H, lens = 64, 8, [40, 64, 17]
    g = torch.Generator().manual_seed(7)
    d, n = hd * H, sum(lens)
    qkv = torch.randn(n, 3 * d, generator=g)
    qkv[:, :d] *= 6.0                                          # scores of +-50: most rows are nearly one-hot
    qkv[:, 2 * d:] *= vmag
    qkv64 = qkv.double().requires_grad_(True)
    dO = torch.randn(n, d, generator=g) * gmag
    off = np.concatenate([[0], np.cumsum(lens)])
    outs = []
    for s in range(len(lens)):
        a, b = off[s], off[s + 1]
        q = qkv64[a:b, :d].reshape(-1, H, hd).transpose(0, 1)
        k = qkv64[a:b, d:2 * d].reshape(-1, H, hd).transpose(0, 1)
        v = qkv64[a:b, 2 * d:].reshape(-1, H, hd).transpose(0, 1)
        outs.append((torch.softmax(q @ k.transpose(1, 2) / math.sqrt(hd), -1) @ v).transpose(0, 1).reshape(-1, d))
    torch.cat(outs).backward(dO.double())
    qd, dod, offd = dev(qkv), dev(dO), i32(off)
    o = torch.empty(n, d, device=DEV)
    lse = torch.empty(n, H, device=DEV)
    _lib.check(lib.bigc_mha_core_f32(S(), qd.data_ptr(), 3 * d, qd.data_ptr() + 4 * d, 3 * d, qd.data_ptr() + 8 * d, 3 * d, o.data_ptr(), d,
                                     offd.data_ptr(), offd.data_ptr(), len(lens), max(lens), H, hd, lse.data_ptr()))
    got = {}
    for exact in (0, 1):
        lib.bigc_linear_set_exact(exact)
        try:
            dqkv = torch.full((n, 3 * d), float("nan"), device=DEV)
            delta = torch.empty(n, H, device=DEV)
            _lib.check(lib.bigc_mha_bwd_f32(S(), qd.data_ptr(), 3 * d, qd.data_ptr() + 4 * d, 3 * d, qd.data_ptr() + 8 * d, 3 * d, o.data_ptr(), d,
                                            dod.data_ptr(), d, lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), 3 * d, dqkv.data_ptr() + 4 * d, 3 * d,
                                            dqkv.data_ptr() + 8 * d, 3 * d, offd.data_ptr(), offd.data_ptr(), len(lens), max(lens), max(lens), n, H, hd))
            torch.cuda.synchronize()
        finally:
            lib.bigc_linear_set_exact(0)
        assert not torch.isnan(dqkv).any()
        want = qkv64.grad
        got[exact] = [float((dqkv[:, i * d:(i + 1) * d].cpu().double() - want[:, i * d:(i + 1) * d]).abs().max() / want[:, i * d:(i + 1) * d].abs().max())
                      for i in range(3)]
    # the split kernels are as close to the f64 truth as the f32-MFMA kernels (which carry the rounding of the forward's o / lse too)
    for i in range(3):
        assert got[0][i] < max(2.0 * got[1][i], 2e-5), (gmag, vmag, got)
        assert got[0][i] < 2e-4, (gmag, vmag, got)


@pytest.mark.parametrize("nq,half,d,ns", [(192, 256, 512, [64, 5]), (12, 8, 16, [6, 9, 140]), (40, 40, 72, [33, 70, 1]), (20, 8, 10, [3, 66])])
def test_role_attention_backward_chain(nq, half, d, ns):
    """d att -> d scores -> d p2a / d e2a, and the values products, against autograd."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(nq + half)
    B, NT = len(ns), sum(ns)
    p2a = torch.randn(B * nq, 2 * half, generator=g).requires_grad_(True)
    e2a = torch.randn(NT, 2 * half, generator=g).requires_grad_(True)
    enco = torch.randn(NT, d, generator=g).requires_grad_(True)
    off = np.concatenate([[0], np.cumsum(ns)])
    scale = 1.0 / math.sqrt(d)
    gvals = torch.randn(2, B * nq, d, generator=g)
    gatt_extra = [torch.randn(2, nq, n, generator=g) * 0.1 for n in ns]
    total = 0
    atts = []
    for v in range(B):
        e, p, x = e2a[off[v]:off[v + 1]], p2a[v * nq:(v + 1) * nq], enco[off[v]:off[v + 1]]
        s = torch.stack([p[:, :half] @ e[:, :half].t(), p[:, half:] @ e[:, half:].t()], 0) * scale
        a = torch.softmax(s, 2) * torch.softmax(s, 0)
        atts.append(a)
        vals = torch.stack([a[0] @ x, a[1] @ x], 0)
        total = total + (vals * gvals[:, v * nq:(v + 1) * nq]).sum() + (a * gatt_extra[v]).sum()
    total.backward()
    pd, ed, xd, offd, gvd = dev(p2a.detach()), dev(e2a.detach()), dev(enco.detach()), i32(off), dev(gvals)
    att = torch.empty(2 * nq * NT, device=DEV)
    _lib.check(lib.bigc_role_attention_f32(S(), pd.data_ptr(), 2 * half, ed.data_ptr(), 2 * half, att.data_ptr(), offd.data_ptr(), B, nq, half, scale))
    gatt = torch.full((2 * nq * NT,), float("nan"), device=DEV)
    mt = max(ns)
    _lib.check(lib.bigc_role_small_gemm_f32(S(), 0, None, gvd.data_ptr(), d, xd.data_ptr(), d, gatt.data_ptr(), 0, offd.data_ptr(), B, nq, d, half, mt))
    extra = dev(torch.cat([t.reshape(-1) for t in gatt_extra]))
    _lib.check(lib.bigc_axpy_f32(S(), gatt.data_ptr(), extra.data_ptr(), gatt.numel()))
    genco = torch.zeros(NT, d, device=DEV)
    _lib.check(lib.bigc_role_small_gemm_f32(S(), 1, att.data_ptr(), gvd.data_ptr(), d, None, d, genco.data_ptr(), d, offd.data_ptr(), B, nq, d, half, mt))
    ds = torch.full((2 * nq * NT,), float("nan"), device=DEV)
    _lib.check(lib.bigc_role_attention_bwd_f32(S(), pd.data_ptr(), 2 * half, ed.data_ptr(), 2 * half, gatt.data_ptr(), ds.data_ptr(), offd.data_ptr(),
                                               B, nq, half, scale))
    gp = torch.full((B * nq, 2 * half), float("nan"), device=DEV)
    ge = torch.full((NT, 2 * half), float("nan"), device=DEV)
    _lib.check(lib.bigc_role_small_gemm_f32(S(), 2, ds.data_ptr(), None, 0, ed.data_ptr(), 2 * half, gp.data_ptr(), 2 * half, offd.data_ptr(), B, nq, d, half, mt))
    _lib.check(lib.bigc_role_small_gemm_f32(S(), 3, ds.data_ptr(), pd.data_ptr(), 2 * half, None, 0, ge.data_ptr(), 2 * half, offd.data_ptr(), B, nq, d, half, mt))
    torch.cuda.synchronize()
    assert rel(genco, enco.grad) < 2e-5
    assert rel(gp, p2a.grad) < 5e-5
    assert rel(ge, e2a.grad) < 5e-5
    # the paired launches the training step uses (two products per launch): the same bits as the single calls
    gatt2 = torch.full((2 * nq * NT,), float("nan"), device=DEV)
    genco2 = torch.zeros(NT, d, device=DEV)
    _lib.check(lib.bigc_role_small_gemm_pair_f32(S(), 0, att.data_ptr(), gvd.data_ptr(), d, xd.data_ptr(), d, gatt2.data_ptr(), genco2.data_ptr(), d,
                                                 offd.data_ptr(), B, nq, d, half, mt))
    _lib.check(lib.bigc_axpy_f32(S(), gatt2.data_ptr(), extra.data_ptr(), gatt2.numel()))
    gp2 = torch.full((B * nq, 2 * half), float("nan"), device=DEV)
    ge2 = torch.full((NT, 2 * half), float("nan"), device=DEV)
    _lib.check(lib.bigc_role_small_gemm_pair_f32(S(), 1, ds.data_ptr(), pd.data_ptr(), 2 * half, ed.data_ptr(), 2 * half, gp2.data_ptr(), ge2.data_ptr(),
                                                 2 * half, offd.data_ptr(), B, nq, d, half, mt))
    torch.cuda.synchronize()
    assert torch.equal(gatt2, gatt) and torch.equal(genco2, genco) and torch.equal(gp2, gp) and torch.equal(ge2, ge)


def test_pool_conv_gather_periodic_transpose_relu():
    lib = _lib.load()
    g = torch.Generator().manual_seed(8)
    lens_per_vid = [[5, 5, 5], [8, 63, 20, 1], [12, 7]]
    lens = [l for v in lens_per_vid for l in v]
    trk_vid = [i for i, v in enumerate(lens_per_vid) for _ in v]
    row_off = np.concatenate([[0], np.cumsum(lens)])
    tmax = [max(v) for v in lens_per_vid]
    nt, d, P = len(lens), 12, 4
    t_o = [(tmax[v] - 1) // 2 + 1 for v in trk_vid]
    conv_off = np.concatenate([[0], np.cumsum(t_o)])
    ro, co, tv, tm = i32(row_off), i32(conv_off), i32(trk_vid), i32(tmax)
    conv = torch.randn(int(conv_off[-1]), d, generator=g).requires_grad_(True)
    dpool = torch.randn(nt, d * P, generator=g)
    for t in range(nt):
        F.adaptive_max_pool1d(conv[conv_off[t]:conv_off[t + 1]].t()[None], P).reshape(-1).backward(dpool[t], retain_graph=True)
    dconv = torch.full(conv.shape, float("nan"), device=DEV)
    cd, dpd = dev(conv.detach()), dev(dpool)
    _lib.check(lib.bigc_maxpool_bwd_f32(S(), cd.data_ptr(), co.data_ptr(), dpd.data_ptr(), dconv.data_ptr(), nt, d, P))
    torch.cuda.synchronize()
    assert rel(dconv, conv.grad) < 1e-6
    # conv gather backward == scatter-add through the forward gather table
    idx = torch.empty(int(conv_off[-1]), 3, dtype=torch.int32, device=DEV)
    _lib.check(lib.bigc_conv_index(S(), ro.data_ptr(), co.data_ptr(), tv.data_ptr(), tm.data_ptr(), idx.data_ptr(), nt))
    dc = torch.randn(int(conv_off[-1]), d, generator=g)
    G = torch.full((int(row_off[-1]), 3, d), float("nan"), device=DEV)
    dcd = dev(dc)
    _lib.check(lib.bigc_conv_gather_bwd_f32(S(), dcd.data_ptr(), ro.data_ptr(), co.data_ptr(), tv.data_ptr(), tm.data_ptr(), G.data_ptr(), nt, d))
    torch.cuda.synchronize()
    want = torch.zeros(int(row_off[-1]), 3, d)
    ih = idx.cpu().long()
    for j in range(ih.shape[0]):
        for k in range(3):
            if ih[j, k] >= 0:
                want[ih[j, k], k] += dc[j]
    assert rel(G, want) < 1e-6
    # periodic reduction, transpose, relu mask
    x = torch.randn(36, 16, generator=g)
    out0 = torch.randn(12, 16, generator=g)
    o, xd = dev(out0), dev(x)
    _lib.check(lib.bigc_reduce_periodic_f32(S(), xd.data_ptr(), o.data_ptr(), 36, 16, 12))
    a = torch.randn(37, 133, generator=g)
    at, ad = torch.empty(133, 37, device=DEV), dev(a)
    _lib.check(lib.bigc_transpose_f32(S(), ad.data_ptr(), at.data_ptr(), 37, 133))
    y = torch.randn(40, 8, generator=g)
    dy = torch.randn(40, 8, generator=g)
    dyd, yd = dev(dy), dev(torch.relu(y))
    _lib.check(lib.bigc_relu_bwd_f32(S(), dyd.data_ptr(), yd.data_ptr(), 320))
    torch.cuda.synchronize()
    assert rel(o, out0 + x.reshape(3, 12, 16).sum(0)) < 1e-6
    assert torch.equal(at.cpu(), a.t())
    assert torch.equal(dyd.cpu(), dy * (y > 0))


def build_train_model(cfg, variant, sd):
    from models import BIG_C_vidvrd, BIG_C_vidor
    d = tempfile.mkdtemp()
    cfg = dict(cfg)
    np.save(os.path.join(d, "bias.npy"), sd["bias_matrix"].numpy())
    cfg["bias_matrix_path"] = os.path.join(d, "bias.npy")
    if "EntiNameEmb" in sd:
        np.save(os.path.join(d, "emb.npy"), sd["EntiNameEmb"].numpy())
        cfg["EntiNameEmb_path"] = os.path.join(d, "emb.npy")
    model = (BIG_C_vidvrd if variant == "vidvrd" else BIG_C_vidor)(cfg, is_train=True)
    model.load_state_dict(sd, strict=True)
    return model.to(DEV).eval()        # parity with the reference is defined with dropout off (is_train selects the training PATH)


@pytest.mark.parametrize("name", list(cases.TRAIN_CASES))
def test_train_step_vs_reference_golden(name):
    """loss terms, Hungarian indices and the gradient of every parameter vs the real reference."""
    g = np.load(os.path.join(cases.GOLDEN_DIR, name + ".npz"))
    cfg, variant, sd, props, gts = cases.make_train_case(name)
    model = build_train_model(cfg, variant, sd)
    total, terms = model(props, gts)
    assert set(terms) == {"cls_pos", "cls_neg", "adj"}
    total.backward()
    torch.cuda.synchronize()
    assert abs(float(total) - float(g["total"])) < 2e-4 * abs(float(g["total"]))
    for k, v in terms.items():
        assert abs(float(v) - float(g["loss_" + k])) < 2e-4 * max(1.0, abs(float(g["loss_" + k]))), k
    for i, (iq, ig) in enumerate(model.last_match_indices):
        assert np.array_equal(np.asarray(iq), g["match%d_query" % i]) and np.array_equal(np.asarray(ig), g["match%d_gt" % i])
    worst = {}
    for k, p in model.named_parameters():
        want = float(g["gnorm_" + k])
        if not p.requires_grad:
            assert p.grad is None
            continue
        assert p.grad is not None, k
        got = float(p.grad.norm())
        worst[k] = abs(got - want) / max(want, 1e-8)
        if "grad_" + k in g.files:
            ref = g["grad_" + k]
            worst[k] = max(worst[k], float(np.abs(p.grad.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-12)))
    # Typical deviation is ~1e-5.  A ReLU whose pre-activation lies within f32 rounding of zero may take the other branch
    # than in the reference run (either is a valid subgradient); one such unit changes the gradient of its own layer by
    # ~2e-3 and everything upstream of it by a few 1e-4 (measured: decoder layer 3's fc2.0 against the exact-GEMM mode).
    # So: every parameter within 3e-3, and half of them within 1e-4.
    bad = {k: v for k, v in worst.items() if v > 3e-3}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
    assert float(np.median(list(worst.values()))) < 1e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:10]


def test_train_step_then_optimizer_like_the_reference_loop():
    """tools/train_vidvrd.py:151-164 unchanged: zero_grad, forward, mean, backward, clip, Adam step."""
    cfg, variant, sd, props, gts = cases.make_train_case("bigc_train_tiny")
    model = build_train_model(cfg, variant, sd)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(3):
        _, _, _, _, gts = cases.make_train_case("bigc_train_tiny")     # the forward mutates the GT durations (reference quirk)
        opt.zero_grad()
        loss, terms = model(props, gts)
        loss = loss.mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=5, norm_type=2)
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("name,fused", [("bigc_train_tiny", False), ("bigc_train_tiny", True), ("bigc_train_v", True)])
def test_gradients_of_later_optimizer_steps_follow_the_updated_weights(name, fused):
    """Every buffer DERIVED from a weight (split forms, transposed copies, the backward's repack of the conv weight) must be
    refreshed after each optimizer step: the gradients of the 2nd and 3rd step, taken at the weights Adam produced, are compared
    with the oracle's autograd AT THOSE WEIGHTS (a derived buffer left at the previous step's values shows up in the gradients
    of everything below the conv: fc_bbox2enti.*, fc_feat2enti.*).  ``fused``: torch's fused Adam -- what tools/ and bench.py use --
    updates the parameters WITHOUT bumping their version counters, so the step cannot rely on them to notice new weights."""
    cfg, variant, sd, props, _ = cases.make_train_case(name)
    model = build_train_model(cfg, variant, sd)
    # a large step (stale weights would be far from the current ones), but not so large that the exp2-dims model diverges
    opt = torch.optim.Adam(model.parameters(), lr=3e-3 if name == "bigc_train_tiny" else 3e-4, fused=fused)
    for step in range(3):
        gts = cases.make_train_case(name)[4]
        opt.zero_grad()
        loss, _ = model(props, gts)
        loss.backward()
        torch.cuda.synchronize()
        if step > 0:
            sdg = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
            want, _, _ = T.forward_train(sdg, cfg, props, cases.make_train_case(name)[4], variant)
            want.backward()
            assert abs(float(loss) - float(want)) < 2e-4 * abs(float(want))
            worst = {}
            # (relative to the parameter's own largest gradient entry, or -- for a parameter whose gradient all but vanished, e.g.
            # fc_enti2att.bias under a saturated attention -- to 1e-4 of the largest entry of any gradient of the step)
            gmax = max(float(sdg[k].grad.abs().max()) for k, p in model.named_parameters() if p.grad is not None)
            for k, p in model.named_parameters():
                if p.grad is None:
                    continue
                ref = sdg[k].grad
                worst[k] = float((p.grad.cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-4 * gmax))
            # A stale derived buffer is wrong by 2e-2 .. 2e-1 in MANY parameters below it (measured: conv 0.19, fc_feat2enti.* 0.02-0.04).
            # What the bound must let through is ONE ReLU unit whose pre-activation lies within f32 rounding of zero and takes the
            # other branch than in the oracle (either is a valid subgradient): it moves the gradient of its own layer by up to 6e-3
            # and nothing else above 3e-3 (seen at step 3 of the exp2-dims case in 4 of 10 runs -- the weights Adam produces differ in
            # their last bits from run to run, the bias gradients being float atomics -- always as fc_feat2enti.2.weight 5.75e-3).
            bad = {k: v for k, v in worst.items() if v > 3e-3}
            assert len(bad) <= 1 and all(v < 1.5e-2 for v in bad.values()), (step, sorted(bad.items(), key=lambda kv: -kv[1])[:10])
            front = sorted(v for k, v in worst.items() if k.startswith(("fc_bbox2enti", "fc_feat2enti")))
            assert len(front) >= 4 and front[-2] < 3e-3 and front[-1] < 1.5e-2, (step, front)
            assert float(np.median(list(worst.values()))) < 2e-4, (step, sorted(worst.items(), key=lambda kv: -kv[1])[:10])
        opt.step()


def test_eval_forward_after_a_fused_optimizer_step_uses_the_new_weights():
    """``Adam(fused=True)`` does not bump the parameters' version counters, and an EVAL forward on the same module
    (in-process validation: flip ``is_train``) does not pass ``force``.  The training backward now marks the engine's derived weight
    buffers dirty, so the validation forward after the step must equal the forward of a fresh model built from the new state_dict."""
    cfg, variant, sd, props, _ = cases.make_train_case("bigc_train_tiny")
    model = build_train_model(cfg, variant, sd)
    opt = torch.optim.Adam(model.parameters(), lr=3e-3, fused=True)
    model.is_train = False
    before = model(props, topk=3)                       # the engine now holds the split copies of the INITIAL weights
    model.is_train = True
    loss, _ = model(props, cases.make_train_case("bigc_train_tiny")[4])
    loss.backward()
    opt.step()
    model.is_train = False
    after = model(props, topk=3)
    fresh = build_train_model(cfg, variant, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    fresh.is_train = False
    want = fresh(props, topk=3)
    changed = False
    for a, w, b in zip(after, want, before):
        assert (a is None) == (w is None)
        if a is not None:
            assert all(torch.equal(x, y) for x, y in zip(a, w))
            changed = changed or b is None or a[1].shape != b[1].shape or not torch.equal(a[1], b[1])
    assert changed                                       # (the step did move the scores: the check above is not vacuous)


@pytest.mark.parametrize("name", ["bigc_train_tiny", "bigc_train_v"])
def test_train_step_exact_gemm_mode_every_gradient_within_1e_4(name):
    """The same golden with every GEMM on the exact f32 MFMA (bigc_linear_set_exact(1)): no ReLU sits on the other side of zero
    than in the reference run, and EVERY parameter gradient agrees within 1e-4 (the 3e-3 slack of the default-mode test above is
    that single branch flip, not an error of the backward)."""
    from bigc_hip import _lib
    lib = _lib.load()
    g = np.load(os.path.join(cases.GOLDEN_DIR, name + ".npz"))
    cfg, variant, sd, props, gts = cases.make_train_case(name)
    model = build_train_model(cfg, variant, sd)
    prev = lib.bigc_linear_set_exact(1)
    try:
        total, terms = model(props, gts)
        total.backward()
        torch.cuda.synchronize()
    finally:
        lib.bigc_linear_set_exact(prev)
    assert abs(float(total) - float(g["total"])) < 1e-4 * abs(float(g["total"]))
    worst = {}
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        want = float(g["gnorm_" + k])
        worst[k] = abs(float(p.grad.norm()) - want) / max(want, 1e-8)
        if "grad_" + k in g.files:
            ref = g["grad_" + k]
            worst[k] = max(worst[k], float(np.abs(p.grad.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-12)))
    bad = {k: v for k, v in worst.items() if v > 1e-4}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


def test_two_forwards_before_one_backward_accumulate_like_autograd():
    """gradient accumulation over micro-batches: loss_a + loss_b built from two forwards, ONE backward -- every step keeps its
    own saved activations, so the result is the sum of the two separately computed gradients (the reference's autograd graph
    supports this pattern)"""
    cfg, variant, sd, props, gts = cases.make_train_case("bigc_train_tiny")
    model = build_train_model(cfg, variant, sd)

    def fresh_gts():
        return cases.make_train_case("bigc_train_tiny")[4]       # the forward mutates the GT durations (reference quirk)

    grads = []
    for sel in ([0], [1]):
        g = fresh_gts()
        model.zero_grad()
        loss, _ = model([props[i] for i in sel], [g[i] for i in sel])
        loss.backward()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    g = fresh_gts()
    model.zero_grad()
    la, _ = model([props[0]], [g[0]])
    lb, _ = model([props[1]], [g[1]])              # second forward BEFORE the first backward
    (la + lb).backward()
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        want = grads[0][k] + grads[1][k]
        assert float((p.grad - want).abs().max()) <= 1e-5 * max(float(want.abs().max()), 1e-6), k


def test_gradients_are_bit_reproducible_run_to_run():
    """The split-M weight-gradient GEMM sums its partial tiles in split order (csrc/backward.hip: dw_reduce_kernel, or the
    last-arriving workgroup under BIGC_DW_REDUCE=inline), and since round 6 no reduction of the step goes through float atomics: two runs
    of the same step give identical bits for EVERY gradient and the loss (five runs, three models: tests/test_hip_determinism.py)."""
    runs = []
    for _ in range(2):
        cfg, variant, sd, props, gts = cases.make_train_case("bigc_train_v")
        model = build_train_model(cfg, variant, sd)
        model.eval()                                  # no dropout
        total, _ = model(props, gts)
        total.backward()
        torch.cuda.synchronize()
        runs.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        runs[-1]["loss"] = total.detach().clone()
    diff = [k for k in runs[0] if not torch.equal(runs[0][k], runs[1][k])]
    assert not diff, diff[:10]


def test_linear_dw_in_kernel_reduction_equals_the_two_launch_form():
    """split-M partial tiles: summed by the last-arriving workgroup (BIGC_DW_REDUCE=inline) == summed by the separate reduce kernel
    (default), bit for bit"""
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, "vidsgg-big_amd")
from bigc_hip import _lib
lib = _lib.load()
torch.manual_seed(5)
M, N, K = 1536, 512, 512
dY = torch.randn(M, N, device="cuda:0"); X = torch.randn(M, K, device="cuda:0")
dW = torch.zeros(N, K, device="cuda:0"); db = torch.zeros(N, device="cuda:0")
for _ in range(2):
    _lib.check(lib.bigc_linear_dw_f32(_lib.current_stream(), dY.data_ptr(), N, X.data_ptr(), K, None, 1, K, dW.data_ptr(), K, db.data_ptr(), M, N, K))
torch.cuda.synchronize()
import hashlib
print("HASH", hashlib.sha1(dW.cpu().numpy().tobytes()).hexdigest(), float((dW - 2 * dY.t() @ X).abs().max() / (dY.t() @ X).abs().max()))
'''
    out = {}
    for mode in ("inline", "kernel"):
        env = dict(os.environ, BIGC_DW_REDUCE=mode)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("HASH")][0].split()
        out[mode] = line[1]
        assert float(line[2]) < 1e-5
    assert out["inline"] == out["kernel"]
