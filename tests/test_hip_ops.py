"""GPU parity of every kernel entry point of the C-ABI library against the CPU oracle /
plain torch fp32 CPU ops on the same seeded inputs.  Run with ``-m gpu`` on an MI355X."""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from bigc_hip import _lib
from oracle import bigc_oracle as O

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return _lib.load()


def S():
    return _lib.current_stream()


def dev(t):
    return t.to(DEV).contiguous()


def i32(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.int32).to(DEV)


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def run_linear(lib, A, W, bias=None, res=None, act=0, idx=None, nseg=1, seg_len=0, M=None):
    M = A.shape[0] if M is None else M
    N, K = W.shape
    C = torch.full((M, N), float("nan"), device=DEV)
    _lib.check(lib.bigc_linear_f32(S(), A.data_ptr(), A.stride(0), _lib.ptr(idx), nseg, seg_len, W.data_ptr(), W.stride(0),
                                   _lib.ptr(bias), _lib.ptr(res), 0 if res is None else res.stride(0), C.data_ptr(), N, M, N, K, act))
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize("M,N,K", [(1, 16, 8), (37, 133, 3160), (300, 512, 2048), (129, 257, 36), (64, 1536, 512),
                                   (1000, 512, 8), (257, 128, 300)])
@pytest.mark.parametrize("act,use_res", [(0, False), (1, True)])
def test_linear(lib, M, N, K, act, use_res):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) if use_res else None
    want = F.linear(A, W, b)
    if act:
        want = torch.relu(want)
    if use_res:
        want = want + res
    got = run_linear(lib, dev(A), dev(W), dev(b), None if res is None else dev(res), act)
    assert not torch.isnan(got).any()
    assert rel_err(got, want) < 2e-6


def test_linear_strided_output_and_input(lib):
    """column-offset writes into a wider buffer (the [bbox|visual] concat) and lda > K reads."""
    g = torch.Generator().manual_seed(5)
    A = torch.randn(200, 96, generator=g)
    W = torch.randn(40, 64, generator=g)
    Ad, Wd = dev(A), dev(W)
    out = torch.zeros(200, 100, device=DEV)
    view = out[:, 50:90]
    _lib.check(lib.bigc_linear_f32(S(), Ad.data_ptr(), 96, None, 1, 64, Wd.data_ptr(), 64, None, None, 0,
                                   view.data_ptr(), 100, 200, 40, 64, 0))
    torch.cuda.synchronize()
    want = F.linear(A[:, :64], W)
    assert rel_err(out[:, 50:90], want) < 2e-6
    assert float(out[:, :50].abs().max()) == 0 and float(out[:, 90:].abs().max()) == 0


def test_linear_gather_segments(lib):
    """segmented row gather = the implicit GEMM of the k3/s2 conv, with -1 (zero padding) rows."""
    g = torch.Generator().manual_seed(9)
    src = torch.randn(50, 64, generator=g)
    M, nseg = 333, 3
    idx = torch.randint(-1, 50, (M, nseg), generator=g)
    W = torch.randn(48, nseg * 64, generator=g) / 8
    b = torch.randn(48, generator=g)
    Ag = torch.zeros(M, nseg, 64)
    for s in range(nseg):
        ok = idx[:, s] >= 0
        Ag[ok, s] = src[idx[ok, s]]
    want = F.linear(Ag.reshape(M, -1), W, b)
    got = run_linear(lib, dev(src), dev(W), dev(b), idx=i32(idx.numpy()), nseg=nseg, seg_len=64, M=M)
    assert rel_err(got, want) < 2e-6
    # plain row gather (nseg == 1), K not a multiple of the 32-wide slab
    src2 = torch.randn(20, 40, generator=g)
    idx1 = torch.randint(0, 20, (77, 1), generator=g)
    W2 = torch.randn(10, 40, generator=g)
    got = run_linear(lib, dev(src2), dev(W2), idx=i32(idx1.numpy()), nseg=1, seg_len=40, M=77)
    assert rel_err(got, F.linear(src2[idx1[:, 0]], W2)) < 2e-6


@pytest.mark.parametrize("M,N,K,act,use_res,gather", [
    (1536, 512, 512, 0, True, False),       # the decoder GEMM of an 8-video training step (48 tiles: splits)
    (1536, 512, 1024, 1, False, False),     # ReLU applied by the reduction
    (515, 512, 3160, 0, True, False),       # K tail (3160 = 98 slabs + 24), M not a tile multiple
    (130, 260, 780, 1, True, False),        # N, M, K all ragged (K and N multiples of 4 as the kernels need, K not of the 32-wide slab)
    (333, 48, 576, 0, False, True),         # gathered operand (3 segments of 192)
    (64, 1536, 512, 0, False, False),
])
def test_linear_split_k_equals_the_unsplit_form_up_to_reassociation(lib, M, N, K, act, use_res, gather):
    """the split-K path (training steps only: bigc_linear_splitk_allow) had no test of its own.  Same operands through
    bigc_linear_f32 with and without the scope: equal up to the reassociation of K (partial tiles added in split order), bias / activation /
    residual applied once by the reduction; and both within 2e-6 of torch."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) if use_res else None
    if gather:
        nseg, seg = 3, K // 3
        src = torch.randn(90, seg, generator=g)
        idx = torch.randint(-1, 90, (M, nseg), generator=g)
        Ag = torch.zeros(M, nseg, seg)
        for s_ in range(nseg):
            ok = idx[:, s_] >= 0
            Ag[ok, s_] = src[idx[ok, s_]]
        A_full = Ag.reshape(M, -1)
        kw = dict(idx=i32(idx.numpy()), nseg=nseg, seg_len=seg, M=M)
        A_dev = dev(src)
    else:
        A_full = torch.randn(M, K, generator=g)
        kw = {}
        A_dev = dev(A_full)
    want = F.linear(A_full, W, b)
    if act:
        want = torch.relu(want)
    if use_res:
        want = want + res                       # (the residual is added behind the activation, as in test_linear)
    plain = run_linear(lib, A_dev, dev(W), dev(b), None if res is None else dev(res), act, **kw)
    prev = lib.bigc_linear_splitk_allow(1)
    try:
        split = run_linear(lib, A_dev, dev(W), dev(b), None if res is None else dev(res), act, **kw)
    finally:
        lib.bigc_linear_splitk_allow(prev)
    assert prev == 0
    assert rel_err(plain, want) < 2e-6 and rel_err(split, want) < 2e-6
    assert rel_err(split, plain) < 1e-6
    assert not torch.isnan(split).any()


def test_linear_rejects_bad_shapes(lib):
    A = torch.zeros(4, 6, device=DEV)
    W = torch.zeros(4, 6, device=DEV)
    C = torch.zeros(4, 4, device=DEV)
    with pytest.raises(AssertionError):
        _lib.check(lib.bigc_linear_f32(S(), A.data_ptr(), 6, None, 1, 6, W.data_ptr(), 6, None, None, 0, C.data_ptr(), 4, 4, 4, 6, 0))


@pytest.mark.parametrize("rows,d", [(1, 16), (7, 128), (1000, 512), (33, 2048), (1001, 128), (49, 64), (35, 256), (3, 512)])
def test_layernorm(lib, rows, d):
    g = torch.Generator().manual_seed(rows + d)
    x = torch.randn(rows, d, generator=g) * 3 + 1
    w = torch.randn(d, generator=g)
    b = torch.randn(d, generator=g)
    y = torch.empty(rows, d, device=DEV)
    xd, wd, bd = dev(x), dev(w), dev(b)
    _lib.check(lib.bigc_layernorm_f32(S(), xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), rows, d, 1e-5))
    torch.cuda.synchronize()
    assert rel_err(y, F.layer_norm(x, (d,), w, b, 1e-5)) < 2e-6


@pytest.mark.parametrize("rows,d,period,both", [(1000, 512, 192, True), (77, 512, 12, False), (50, 48, 7, True), (130, 128, 130, True)])
def test_layernorm_add_is_layernorm_then_add_rows(lib, rows, d, period, both):
    """fused LayerNorm + broadcast add: bit-identical to bigc_layernorm_f32 followed by bigc_add_rows_f32"""
    g = torch.Generator().manual_seed(rows + d + period)
    x, w, b, p = dev(torch.randn(rows, d, generator=g) * 2), dev(torch.randn(d, generator=g)), dev(torch.randn(d, generator=g)), \
        dev(torch.randn(period, d, generator=g))
    ln = torch.empty(rows, d, device=DEV)
    want = torch.empty(rows, d, device=DEV)
    _lib.check(lib.bigc_layernorm_f32(S(), x.data_ptr(), w.data_ptr(), b.data_ptr(), ln.data_ptr(), rows, d, 1e-5))
    _lib.check(lib.bigc_add_rows_f32(S(), ln.data_ptr(), p.data_ptr(), want.data_ptr(), rows, d, period))
    y = torch.full((rows, d), float("nan"), device=DEV)
    plain = torch.full((rows, d), float("nan"), device=DEV)
    _lib.check(lib.bigc_layernorm_add_f32(S(), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, d, 1e-5, p.data_ptr(), period,
                                          plain.data_ptr() if both else None))
    torch.cuda.synchronize()
    assert torch.equal(y, want)
    if both:
        assert torch.equal(plain, ln)


@pytest.mark.parametrize("rows,d,period", [(1000, 512, 192), (77, 128, 0), (3, 64, 2), (257, 256, 0)])
def test_layernorm_of_a_sum_is_the_sum_then_layernorm(lib, rows, d, period):
    """bigc_layernorm_res_f32 (the residual added by the LayerNorm): bit-identical to an f32 addition followed by bigc_layernorm_add_f32 /
    bigc_layernorm_hr8_f32, also in place (pre aliases y)"""
    g = torch.Generator().manual_seed(rows + d + period)
    x, pre, w, b = dev(torch.randn(rows, d, generator=g) * 2), dev(torch.randn(rows, d, generator=g)), dev(torch.randn(d, generator=g)), dev(torch.randn(d, generator=g))
    summed = x + pre
    if period:
        p = dev(torch.randn(period, d, generator=g))
        want = torch.empty(rows, d, device=DEV)
        _lib.check(lib.bigc_layernorm_add_f32(S(), summed.data_ptr(), w.data_ptr(), b.data_ptr(), want.data_ptr(), rows, d, 1e-5, p.data_ptr(), period, None))
        y = pre.clone()                                  # in place: the residual is overwritten by the result
        _lib.check(lib.bigc_layernorm_res_f32(S(), x.data_ptr(), y.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, 0, rows, d, 1e-5,
                                              p.data_ptr(), period, None))
        torch.cuda.synchronize()
        assert torch.equal(y, want)
        return
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    want, want_s = torch.empty(rows, d, device=DEV), torch.zeros(rows, 4 * d, dtype=torch.uint8, device=DEV)
    _lib.check(lib.bigc_layernorm_hr8_f32(S(), summed.data_ptr(), w.data_ptr(), b.data_ptr(), want.data_ptr(), want_s.data_ptr(), 4 * d, rows, d, 1e-5, flag.data_ptr()))
    y, ys = torch.empty(rows, d, device=DEV), torch.zeros(rows, 4 * d, dtype=torch.uint8, device=DEV)
    _lib.check(lib.bigc_layernorm_res_f32(S(), x.data_ptr(), pre.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), ys.data_ptr(), 4 * d, rows, d, 1e-5,
                                          None, 0, flag.data_ptr()))
    plain = torch.empty(rows, d, device=DEV)
    _lib.check(lib.bigc_layernorm_res_f32(S(), x.data_ptr(), pre.data_ptr(), w.data_ptr(), b.data_ptr(), plain.data_ptr(), None, 0, rows, d, 1e-5, None, 0, None))
    torch.cuda.synchronize()
    assert torch.equal(y, want) and torch.equal(ys, want_s) and torch.equal(plain, want)


def test_add_rows(lib):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(36, 16, generator=g)
    p = torch.randn(12, 16, generator=g)
    y = torch.empty(36, 16, device=DEV)
    xd, pd = dev(x), dev(p)
    _lib.check(lib.bigc_add_rows_f32(S(), xd.data_ptr(), pd.data_ptr(), y.data_ptr(), 36, 16, 12))
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), x + p.repeat(3, 1))
    _lib.check(lib.bigc_add_rows_f32(S(), None, pd.data_ptr(), y.data_ptr(), 36, 16, 12))
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), p.repeat(3, 1))


@pytest.mark.parametrize("hd,H,lens", [(64, 8, [64]), (64, 8, [1, 33, 180, 64]), (8, 2, [6, 5, 9]), (16, 8, [3, 120, 400]),
                                       (32, 4, [31, 32, 65])])
def test_mha_core(lib, hd, H, lens):
    """ragged self-attention core vs softmax(q k^T / sqrt(hd)) v per sequence and head."""
    g = torch.Generator().manual_seed(hd + len(lens))
    d = hd * H
    n = sum(lens)
    qkv = torch.randn(n, 3 * d, generator=g)
    off = np.concatenate([[0], np.cumsum(lens)])
    want = torch.empty(n, d)
    for s in range(len(lens)):
        a, b = off[s], off[s + 1]
        q = qkv[a:b, :d].reshape(-1, H, hd).transpose(0, 1)
        k = qkv[a:b, d:2 * d].reshape(-1, H, hd).transpose(0, 1)
        v = qkv[a:b, 2 * d:].reshape(-1, H, hd).transpose(0, 1)
        att = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(hd), -1)
        want[a:b] = (att @ v).transpose(0, 1).reshape(-1, d)
    qd = dev(qkv)
    o = torch.full((n, d), float("nan"), device=DEV)
    offd = i32(off)
    _lib.check(lib.bigc_mha_core_f32(S(), qd.data_ptr(), 3 * d, qd.data_ptr() + 4 * d, 3 * d, qd.data_ptr() + 8 * d, 3 * d,
                                     o.data_ptr(), d, offd.data_ptr(), offd.data_ptr(), len(lens), max(lens), H, hd, None))
    torch.cuda.synchronize()
    assert not torch.isnan(o).any()
    assert rel_err(o, want) < 5e-6


def test_mha_core_hd16_sequence_does_not_depend_on_its_launch(lib):
    """head_dim 16 (the grounding scorer's attention): a sequence's output must not depend on what else is in the launch -- the launcher
    picks the waves per workgroup from the longest sequence and, for long launches, from the NUMBER of sequences (eight waves per
    sequence for small launches, four above 96 sequences of 8 heads), and walks the keys in chunks whose length is fixed per class.  The
    same three sequences (200, 130 and 90 keys) alone, and as the first three of 120: identical bits."""
    hd, H = 16, 8
    d = hd * H
    g = torch.Generator().manual_seed(77)
    base = [200, 130, 90]
    many = base + [int(v) for v in torch.randint(129, 380, (117,), generator=g)]
    qkv = torch.randn(sum(many), 3 * d, generator=g)
    qd = dev(qkv)

    def run(lens):
        n = sum(lens)
        off = i32(np.concatenate([[0], np.cumsum(lens)]))
        o = torch.full((n, d), float("nan"), device=DEV)
        _lib.check(lib.bigc_mha_core_f32(S(), qd.data_ptr(), 3 * d, qd.data_ptr() + 4 * d, 3 * d, qd.data_ptr() + 8 * d, 3 * d,
                                         o.data_ptr(), d, off.data_ptr(), off.data_ptr(), len(lens), max(lens), H, hd, None))
        torch.cuda.synchronize()
        return o
    few, lots = run(base), run(many)
    assert not torch.isnan(lots).any()
    assert torch.equal(few, lots[:sum(base)])
    # ... and a short sequence (<= 128 keys: one chunk in every class) next to long ones equals the same sequence in a launch of short ones
    short_alone = run([90])            # rows 0 .. 89 of qkv = the first 90 rows of the 200-key sequence: another sequence, same kernel class check
    mixed = run([90, 200])
    assert torch.equal(short_alone, mixed[:90])


@pytest.mark.parametrize("lens", [[40, 64], [150, 20]])
def test_mha_core_out_of_range_falls_back_to_f32(lib, lens):
    """head_dim 64 runs on split f16 operands; a value beyond the f16 range (here 1e5 in v, 4e4 in k) must hand the
    launch to the f32-MFMA kernel and give the same result as the exact mode."""
    g = torch.Generator().manual_seed(5)
    H, hd = 2, 64
    d = H * hd
    n = sum(lens)
    qkv = torch.randn(n, 3 * d, generator=g)
    qkv[3, 2 * d + 7] = 1e5
    qkv[n - 2, d + 70] = -4e4
    off = np.concatenate([[0], np.cumsum(lens)])
    want = torch.empty(n, d, dtype=torch.float64)
    for s in range(len(lens)):
        a, b = off[s], off[s + 1]
        q, k, v = (qkv[a:b, i * d:(i + 1) * d].double().reshape(-1, H, hd).transpose(0, 1) for i in range(3))
        want[a:b] = (torch.softmax(q @ k.transpose(1, 2) / math.sqrt(hd), -1) @ v).transpose(0, 1).reshape(-1, d)
    qd, offd = dev(qkv), i32(off)
    outs = []
    for exact in (0, 1):
        prev = lib.bigc_linear_set_exact(exact)
        try:
            o = torch.full((n, d), float("nan"), device=DEV)
            _lib.check(lib.bigc_mha_core_f32(S(), qd.data_ptr(), 3 * d, qd.data_ptr() + 4 * d, 3 * d, qd.data_ptr() + 8 * d, 3 * d,
                                             o.data_ptr(), d, offd.data_ptr(), offd.data_ptr(), len(lens), max(lens), H, hd, None))
            torch.cuda.synchronize()
        finally:
            lib.bigc_linear_set_exact(prev)
        outs.append(o.cpu())
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0], want.float()) < 5e-6


def test_mha_core_cross_lengths(lib):
    """different query/key offsets (decoder-style: 12 queries per video against themselves is the same
    call; here queries and keys have different counts)."""
    g = torch.Generator().manual_seed(77)
    H, hd = 2, 8
    d = H * hd
    ql, kl = [5, 40], [70, 3]
    q = torch.randn(sum(ql), d, generator=g)
    k = torch.randn(sum(kl), d, generator=g)
    v = torch.randn(sum(kl), d, generator=g)
    qo, ko = np.concatenate([[0], np.cumsum(ql)]), np.concatenate([[0], np.cumsum(kl)])
    want = torch.empty(sum(ql), d)
    for s in range(2):
        qq = q[qo[s]:qo[s + 1]].reshape(-1, H, hd).transpose(0, 1)
        kk = k[ko[s]:ko[s + 1]].reshape(-1, H, hd).transpose(0, 1)
        vv = v[ko[s]:ko[s + 1]].reshape(-1, H, hd).transpose(0, 1)
        want[qo[s]:qo[s + 1]] = (torch.softmax(qq @ kk.transpose(1, 2) / math.sqrt(hd), -1) @ vv).transpose(0, 1).reshape(-1, d)
    qd, kd, vd = dev(q), dev(k), dev(v)
    o = torch.empty(sum(ql), d, device=DEV)
    qod, kod = i32(qo), i32(ko)
    _lib.check(lib.bigc_mha_core_f32(S(), qd.data_ptr(), d, kd.data_ptr(), d, vd.data_ptr(), d, o.data_ptr(), d,
                                     qod.data_ptr(), kod.data_ptr(), 2, max(ql), H, hd, None))
    torch.cuda.synchronize()
    assert rel_err(o, want) < 5e-6


@pytest.mark.parametrize("nq,half,d,ns", [(192, 256, 512, [64, 1, 180]), (12, 8, 16, [6, 9]), (40, 16, 160, [33]), (192, 256, 512, [64, 1, 37]), (70, 64, 96, [5, 32]),
                                          (200, 128, 320, [40, 64]), (33, 256, 288, [64, 2])])
def test_role_attention_and_values(lib, nq, half, d, ns):
    g = torch.Generator().manual_seed(nq + half)
    B, NT = len(ns), sum(ns)
    p2a = torch.randn(B * nq, 2 * half, generator=g)
    e2a = torch.randn(NT, 2 * half, generator=g)
    enco = torch.randn(NT, d, generator=g)
    off = np.concatenate([[0], np.cumsum(ns)])
    scale = 1.0 / math.sqrt(d)
    att_want, val_want = [], torch.empty(2, B * nq, d)
    for v in range(B):
        e = e2a[off[v]:off[v + 1]]
        p = p2a[v * nq:(v + 1) * nq]
        s = torch.stack([p[:, :half] @ e[:, :half].t(), p[:, half:] @ e[:, half:].t()], 0) * scale
        a = torch.softmax(s, 2) * torch.softmax(s, 0)
        att_want.append(a.reshape(-1))
        for r in range(2):
            val_want[r, v * nq:(v + 1) * nq] = a[r] @ enco[off[v]:off[v + 1]]
    att_want = torch.cat(att_want)
    pd, ed, xd, offd = dev(p2a), dev(e2a), dev(enco), i32(off)
    att = torch.full((2 * nq * NT,), float("nan"), device=DEV)
    _lib.check(lib.bigc_role_attention_f32(S(), pd.data_ptr(), 2 * half, ed.data_ptr(), 2 * half, att.data_ptr(), offd.data_ptr(),
                                           B, nq, half, scale))
    vals = torch.full((2, B * nq, d), float("nan"), device=DEV)
    _lib.check(lib.bigc_role_values_f32(S(), att.data_ptr(), xd.data_ptr(), d, vals.data_ptr(), offd.data_ptr(), B, nq, d))
    torch.cuda.synchronize()
    assert not torch.isnan(att).any() and not torch.isnan(vals).any()
    assert rel_err(att, att_want) < 1e-5
    assert rel_err(vals, val_want) < 1e-5
    if max(ns) <= 64 and half % 64 == 0:        # the one-launch variant of the eval path: same results
        att2 = torch.full((2 * nq * NT,), float("nan"), device=DEV)
        vals2 = torch.full((2, B * nq, d), float("nan"), device=DEV)
        _lib.check(lib.bigc_role_fused_f32(S(), pd.data_ptr(), 2 * half, ed.data_ptr(), 2 * half, xd.data_ptr(), d, att2.data_ptr(),
                                           vals2.data_ptr(), offd.data_ptr(), B, nq, max(ns), half, d, scale))
        vals3 = torch.full((2, B * nq, d), float("nan"), device=DEV)
        _lib.check(lib.bigc_role_fused_f32(S(), pd.data_ptr(), 2 * half, ed.data_ptr(), 2 * half, xd.data_ptr(), d, None,
                                           vals3.data_ptr(), offd.data_ptr(), B, nq, max(ns), half, d, scale))
        torch.cuda.synchronize()
        assert rel_err(att2, att_want) < 1e-5 and rel_err(vals2, val_want) < 1e-5
        assert torch.equal(vals2, vals3)


@pytest.mark.parametrize("nq,half,d,ns,exact", [(192, 256, 512, [64, 1, 37], 0), (70, 64, 96, [5, 32], 0), (192, 256, 512, [64, 20], 1),
                                                 (200, 128, 64, [40], 0)])
def test_role_fused_with_projected_values(lib, nq, half, d, ns, exact):
    """bigc_role_fused_proj_f32: values_r = relu(att_r @ proj_r + bias_r) with per-role right-hand rows (the first role-wise
    linear applied to the entity rows), split-operand LDS kernel / f32-MFMA kernel (exact mode, nq > 192)"""
    g = torch.Generator().manual_seed(nq + d + len(ns))
    B, NT = len(ns), sum(ns)
    p2a = torch.randn(B * nq, 2 * half, generator=g)
    e2a = torch.randn(NT, 2 * half, generator=g)
    proj = torch.randn(2, NT, d, generator=g)
    bias = torch.randn(2, d, generator=g)
    off = np.concatenate([[0], np.cumsum(ns)])
    scale = 1.0 / math.sqrt(d)
    att_want, val_want = [], torch.empty(2, B * nq, d)
    for v in range(B):
        e = e2a[off[v]:off[v + 1]]
        p = p2a[v * nq:(v + 1) * nq]
        s = torch.stack([p[:, :half] @ e[:, :half].t(), p[:, half:] @ e[:, half:].t()], 0) * scale
        a = torch.softmax(s, 2) * torch.softmax(s, 0)
        att_want.append(a.reshape(-1))
        for r in range(2):
            val_want[r, v * nq:(v + 1) * nq] = torch.relu(a[r] @ proj[r, off[v]:off[v + 1]] + bias[r])
    att_want = torch.cat(att_want)
    pd, ed, xd, bd, offd = dev(p2a), dev(e2a), dev(proj), dev(bias), i32(off)
    att = torch.full((2 * nq * NT,), float("nan"), device=DEV)
    vals = torch.full((2, B * nq, d), float("nan"), device=DEV)
    side = torch.full((B * nq, 2 * d), float("nan"), device=DEV)             # the two roles side by side in one row
    prev = lib.bigc_linear_set_exact(exact)
    try:
        _lib.check(lib.bigc_role_fused_proj_f32(S(), pd.data_ptr(), 2 * half, ed.data_ptr(), 2 * half, xd.data_ptr(), d, NT * d, bd.data_ptr(),
                                                bd.data_ptr() + 4 * d, 1, att.data_ptr(), vals.data_ptr(), 0, 0, offd.data_ptr(), B, nq, max(ns), half, d,
                                                scale))
        _lib.check(lib.bigc_role_fused_proj_f32(S(), pd.data_ptr(), 2 * half, ed.data_ptr(), 2 * half, xd.data_ptr(), d, NT * d, bd.data_ptr(),
                                                bd.data_ptr() + 4 * d, 1, None, side.data_ptr(), d, 2 * d, offd.data_ptr(), B, nq, max(ns), half, d, scale))
        torch.cuda.synchronize()
    finally:
        lib.bigc_linear_set_exact(prev)
    assert not torch.isnan(vals).any()
    assert rel_err(att, att_want) < 1e-5 and rel_err(vals, val_want) < 1e-5
    assert torch.equal(side[:, :d], vals[0]) and torch.equal(side[:, d:], vals[1])


def test_role_fused_out_of_range_falls_back_to_f32(lib):
    """operands beyond the f16 range (enco 2e5, p2a 5e4): the split-f16 role kernel must hand over to the f32-MFMA kernel and
    give the result of the exact mode"""
    g = torch.Generator().manual_seed(11)
    B, nq, half, d, ns = 2, 40, 64, 64, [50, 9]
    NT = sum(ns)
    p2a = torch.randn(B * nq, 2 * half, generator=g)
    e2a = torch.randn(NT, 2 * half, generator=g)
    enco = torch.randn(NT, d, generator=g)
    enco[3, 5] = 2e5
    p2a[41, 70] = 5e4
    e2a[52, 70] = 1e-3
    off = np.concatenate([[0], np.cumsum(ns)])
    scale = 1.0 / math.sqrt(d)
    pd, ed, xd, offd = dev(p2a), dev(e2a), dev(enco), i32(off)
    outs = []
    for exact in (0, 1):
        prev = lib.bigc_linear_set_exact(exact)
        try:
            att = torch.full((2 * nq * NT,), float("nan"), device=DEV)
            vals = torch.full((2, B * nq, d), float("nan"), device=DEV)
            _lib.check(lib.bigc_role_fused_f32(S(), pd.data_ptr(), 2 * half, ed.data_ptr(), 2 * half, xd.data_ptr(), d, att.data_ptr(),
                                               vals.data_ptr(), offd.data_ptr(), B, nq, max(ns), half, d, scale))
            torch.cuda.synchronize()
        finally:
            lib.bigc_linear_set_exact(prev)
        outs.append((att.cpu(), vals.cpu()))
    assert not torch.isnan(outs[0][1]).any()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def _ragged(lens_per_vid, tmax_extra=0, seed=0, width=24):
    g = torch.Generator().manual_seed(seed)
    lens = [l for v in lens_per_vid for l in v]
    trk_vid = [i for i, v in enumerate(lens_per_vid) for _ in v]
    row_off = np.concatenate([[0], np.cumsum(lens)])
    tmax = [max(v) for v in lens_per_vid]
    boxes = torch.rand(sum(lens), 2, generator=g) * 600
    boxes = torch.cat([boxes, boxes + 10 + torch.rand(sum(lens), 2, generator=g) * 100], 1)
    feats = torch.randn(sum(lens), width, generator=g)
    return lens, trk_vid, row_off, tmax, boxes, feats


def test_bbox_feat_bit_exact(lib):
    lens, trk_vid, row_off, tmax, boxes, _ = _ragged([[5, 9, 1], [300, 2]], seed=1)
    wh = torch.tensor([[1280., 720.], [640., 480.]])
    out = torch.empty(sum(lens), 8, device=DEV)
    bd, ro, tv, whd = dev(boxes), i32(row_off), i32(trk_vid), dev(wh)
    _lib.check(lib.bigc_bbox_feat_f32(S(), bd.data_ptr(), ro.data_ptr(), tv.data_ptr(), whd.data_ptr(), out.data_ptr(), len(lens)))
    torch.cuda.synchronize()
    want = torch.cat([O.bbox8(boxes[row_off[t]:row_off[t + 1]], *wh[trk_vid[t]].tolist()) for t in range(len(lens))])
    assert torch.equal(out.cpu(), want)


def test_conv_index_pool_and_mean(lib):
    lens_per_vid = [[5, 5, 5], [8, 63, 20, 1], [12, 7]]
    lens, trk_vid, row_off, tmax, _, feats = _ragged(lens_per_vid, seed=2)
    nt = len(lens)
    t_o = [(tmax[v] - 1) // 2 + 1 for v in trk_vid]
    conv_off = np.concatenate([[0], np.cumsum(t_o)])
    ro, co, tv, tm = i32(row_off), i32(conv_off), i32(trk_vid), i32(tmax)
    idx = torch.empty(int(conv_off[-1]), 3, dtype=torch.int32, device=DEV)
    _lib.check(lib.bigc_conv_index(S(), ro.data_ptr(), co.data_ptr(), tv.data_ptr(), tm.data_ptr(), idx.data_ptr(), nt))
    torch.cuda.synchronize()
    want = []
    for t in range(nt):
        M = tmax[trk_vid[t]]
        src = O.stretch_index(lens[t], M)
        for j in range(t_o[t]):
            want.append([-1 if (f < 0 or f >= M) else row_off[t] + src[f] for f in (2 * j - 1, 2 * j, 2 * j + 1)])
    assert np.array_equal(idx.cpu().numpy(), np.asarray(want))

    # adaptive max pool, channel-major
    d, P = 20, 4
    g = torch.Generator().manual_seed(4)
    conv = torch.randn(int(conv_off[-1]), d, generator=g)
    pooled = torch.empty(nt, d * P, device=DEV)
    cd = dev(conv)
    _lib.check(lib.bigc_adaptive_maxpool(S(), cd.data_ptr(), co.data_ptr(), pooled.data_ptr(), nt, d, P))
    torch.cuda.synchronize()
    for t in range(nt):
        x = conv[conv_off[t]:conv_off[t + 1]].t()[None]
        assert torch.equal(pooled[t].cpu(), F.adaptive_max_pool1d(x, P).reshape(-1))

    # mean over the stretched frames of a column range
    w, col0 = 12, 8
    out = torch.empty(nt, w, device=DEV)
    fd = dev(feats)
    _lib.check(lib.bigc_stretched_mean(S(), fd.data_ptr(), feats.shape[1], col0, w, ro.data_ptr(), tv.data_ptr(), tm.data_ptr(),
                                       out.data_ptr(), nt))
    torch.cuda.synchronize()
    want = torch.stack([feats[row_off[t]:row_off[t + 1]][torch.from_numpy(O.stretch_index(lens[t], tmax[trk_vid[t]]))][:, col0:col0 + w].mean(0)
                        for t in range(nt)])
    assert rel_err(out, want) < 1e-6
