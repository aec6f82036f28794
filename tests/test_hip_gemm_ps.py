"""GPU tests of the GEMM on split operands (csrc/gemm_ps.hip, entry point bigc_gemm_ps): the three A modes (f32 rows split in
the loop, hr8 rows by DMA, hr8 rows gathered through an index table), f32 and hr8 outputs, bias / ReLU / sigmoid / residual,
ragged M, against a float64 evaluation; accuracy budget = a few f32 round-offs of sum |a||w| (what torch's f32 GEMM delivers)."""
import numpy as np
import pytest
import torch

from bigc_hip import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


ACT_EXP = 3          # PS_ACT_EXP of csrc/linear_internal.h: activations are stored as x * 2^3 = f16 head + f16 residual


def hr8_decode(buf, rows, K, pitch_bytes=None):
    """hr8 bytes (torch uint8 on any device) -> float64 (rows, K): (head + residual) / 2^ACT_EXP"""
    pitch = pitch_bytes or 4 * K
    raw = buf.cpu().numpy().reshape(rows, pitch)[:, :4 * K].copy()
    g = raw.view(np.float16).reshape(rows, K // 8, 2, 8).astype(np.float64)
    return (g[:, :, 0, :] + g[:, :, 1, :]).reshape(rows, K) / 2.0 ** ACT_EXP


def repr_tol(x):
    """representation error bound of the hr8 form of x (float64 array): 2^-23 |x| in the relative regime, 2^-25 / 2^ACT_EXP below"""
    return 2.0 ** -22.9 * np.abs(x) + 2.0 ** -25 / 2.0 ** ACT_EXP


def to_hr8(x):
    rows, K = x.shape
    y = torch.empty((rows, 4 * K), dtype=torch.uint8, device=x.device)
    flag = torch.zeros(1, dtype=torch.int32, device=x.device)
    _lib.check(_lib.load().bigc_split_rows_f32(_lib.current_stream(), x.data_ptr(), K, y.data_ptr(), 4 * K, rows, K, flag.data_ptr()))
    return y, flag


def run(A, a_fmt, W, bias=None, res=None, act=0, idx=None, nseg=1, want_f32=True, want_hr8=True, lda_bytes=None):
    lib = _lib.load()
    M = idx.shape[0] if idx is not None else A.shape[0]
    N, K = W.shape
    C = torch.full((M, N), float("nan"), device=DEV) if want_f32 else None
    Cs = torch.zeros((M, 4 * N), dtype=torch.uint8, device=DEV) if want_hr8 else None
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    if lda_bytes is None:
        lda_bytes = A.shape[1] * (4 if a_fmt == 0 else 1)
    _lib.check(lib.bigc_gemm_ps(_lib.current_stream(), A.data_ptr(), lda_bytes, a_fmt, _lib.ptr(idx), nseg, W.data_ptr(), K, _lib.ptr(bias),
                                _lib.ptr(res), N if res is not None else 0, _lib.ptr(C), N, _lib.ptr(Cs), 4 * N, M, N, K, act, flag.data_ptr()))
    torch.cuda.synchronize()
    return C, Cs, int(flag.item())


def ref64(A, W, bias, res, act):
    y = A.double() @ W.double().t()
    if bias is not None:
        y = y + bias.double()
    if act == 1:
        y = torch.relu(y)
    if act == 2:
        y = torch.sigmoid(y)
    if res is not None:
        y = y + res.double()
    return y


def budget(A, W):
    """sum_k |a||w| per output: the scale f32 round-off is proportional to"""
    return (A.double().abs() @ W.double().abs().t())


def test_split_rows_roundtrip():
    rs = np.random.RandomState(0)
    x = torch.from_numpy((rs.standard_normal((37, 64)) * np.exp(rs.uniform(-12, 7, size=(37, 64)))).astype(np.float32)).to(DEV)
    x[0, :8] = torch.tensor([0.0, 1.0, -1.0, 4095.0, 1e-30, -3e-6, 0.25 / 8, 0.249 / 8])
    y, flag = to_hr8(x)
    back = hr8_decode(y, 37, 64)
    xe = x.double().cpu().numpy()
    assert int(flag.item()) == 0
    assert (np.abs(back - xe) <= repr_tol(xe)).all()
    x[3, 5] = 4096.0
    _, flag = to_hr8(x)
    assert int(flag.item()) == 1
    x[3, 5] = float("nan")
    _, flag = to_hr8(x)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("M,N,K", [(128, 256, 32), (300, 512, 512), (1000, 1536, 512), (77, 512, 2048), (4096 + 5, 1024, 512), (260, 8, 64), (256 * 3 * 256 // 2 + 77, 512, 128),
                                   (192 * 128 - 37, 512, 64)])          # the last one: 192-row tiles (one round of 256 workgroups; 128- and 256-row tilings cost more)
@pytest.mark.parametrize("a_fmt", [0, 1])
def test_gemm_ps_vs_float64(M, N, K, a_fmt):
    rs = np.random.RandomState(M + N + K + a_fmt)
    A = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(DEV)
    res = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32)).to(DEV)
    Ain = A if a_fmt == 0 else to_hr8(A)[0]
    for (b, r, act) in ((bias, None, 1), (None, res, 0), (bias, res, 2), (None, None, 0)):
        C, Cs, flag = run(Ain, a_fmt, W, b, r, act)
        want = ref64(A, W, b, r, act)
        tol = 6 * 2.0 ** -24 * budget(A, W) + 2.0 ** -23 * want.abs() + 1e-7
        err = (C.double() - want).abs()
        assert flag == 0
        assert bool((err <= tol).all()), "f32 output: max err/tol %.2f" % float((err / tol).max())
        back = torch.from_numpy(hr8_decode(Cs, M, N)).to(DEV)
        # the split output is the f32 output rounded to 22+ bits
        assert bool(((back - C.double()).abs() <= 2.0 ** -22.9 * C.double().abs() + 2.0 ** -28).all())
    # outputs one at a time
    C, _, _ = run(Ain, a_fmt, W, bias, None, 1, want_hr8=False)
    _, Cs, _ = run(Ain, a_fmt, W, bias, None, 1, want_f32=False)
    assert bool(((torch.from_numpy(hr8_decode(Cs, M, N)).to(DEV) - C.double()).abs() <= 2.0 ** -22.9 * C.double().abs() + 2.0 ** -28).all())


def test_gemm_ps_result_does_not_depend_on_the_tiling():
    """128-, 192- and 256-row tiles are chosen from the problem size (csrc/gemm_ps.hip: bigc_gemm_ps_launch); every output element is
    ONE accumulator walked over K in the same order in all of them, so the first rows of a big problem equal, bit for bit, the same
    rows computed as a small problem of their own -- a video's decoder rows do not depend on how many videos share the batch."""
    rs = np.random.RandomState(9)
    N, K = 512, 512
    big = {3: 192 * 128, 4: 256 * 3 * 256 // 2, 2: 128 * 64}           # M -> the tiling the launcher's cost rule picks (rows / 64 per tile)
    A = torch.from_numpy(rs.standard_normal((max(big.values()), K)).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(DEV)
    As = to_hr8(A)[0]
    ref, _, _ = run(As[:200], 1, W, bias, None, 1, want_hr8=False)
    for rb, M in big.items():
        C, Cs, flag = run(As[:M], 1, W, bias, None, 1)
        assert flag == 0 and torch.equal(C[:200], ref), rb
        want = ref64(A[:M], W, bias, None, 1)
        assert bool(((C.double() - want).abs() <= 6 * 2.0 ** -24 * budget(A[:M], W) + 2.0 ** -23 * want.abs() + 1e-7).all()), rb


def test_gemm_ps_gather_segments():
    """the conv-as-GEMM form: K = 3 segments, each a source row picked by an index table, -1 = zeros"""
    rs = np.random.RandomState(5)
    n_src, seg, M, N = 700, 256, 900, 512
    src = torch.from_numpy(rs.standard_normal((n_src, seg)).astype(np.float32)).to(DEV)
    idx = torch.from_numpy(rs.randint(-1, n_src, size=(M, 3)).astype(np.int32)).to(DEV)
    idx[5] = -1
    W = torch.from_numpy((rs.standard_normal((N, 3 * seg)) / np.sqrt(3 * seg)).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(DEV)
    zero = torch.zeros((1, seg), device=DEV)
    full = torch.cat([torch.cat([src, zero])[idx[:, s].long()] for s in range(3)], 1)         # index -1 -> the appended zero row
    C, Cs, flag = run(to_hr8(src)[0], 1, W, bias, None, 0, idx=idx, nseg=3)
    want = ref64(full, W, bias, None, 0)
    tol = 6 * 2.0 ** -24 * budget(full, W) + 2.0 ** -23 * want.abs() + 1e-7
    assert flag == 0 and bool(((C.double() - want).abs() <= tol).all())
    assert torch.equal(C[5], bias)


def test_gemm_ps_strided_views():
    """A and the outputs as column blocks of wider buffers (the front-end writes its two branches side by side)"""
    rs = np.random.RandomState(6)
    M, N, K = 500, 512, 512
    wide = torch.from_numpy(rs.standard_normal((M, 2 * K)).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV)
    A = wide[:, K:]
    lib = _lib.load()
    wide_s = torch.empty((M, 8 * K), dtype=torch.uint8, device=DEV)
    _lib.check(lib.bigc_split_rows_f32(_lib.current_stream(), wide.data_ptr(), 2 * K, wide_s.data_ptr(), 8 * K, M, 2 * K, None))
    out = torch.zeros((M, 8 * N), dtype=torch.uint8, device=DEV)            # hr8 output into the right half of a (M, 2N) buffer
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(lib.bigc_gemm_ps(_lib.current_stream(), wide_s.data_ptr() + 4 * K, 8 * K, 1, None, 1, W.data_ptr(), K, None, None, 0,
                                None, 0, out.data_ptr() + 4 * N, 8 * N, M, N, K, 1, flag.data_ptr()))
    torch.cuda.synchronize()
    got = hr8_decode(out, M, 2 * N, 8 * N)
    want = torch.relu(A.double() @ W.double().t()).cpu().numpy()
    assert np.abs(got[:, :N]).max() == 0.0
    assert (np.abs(got[:, N:] - want) <= 6 * 2.0 ** -24 * budget(A, W).cpu().numpy() + 2.0 ** -22 * np.abs(want) + 1e-7).all()


def test_gemm_ps_range_flag():
    rs = np.random.RandomState(7)
    A = torch.from_numpy(rs.standard_normal((256, 64)).astype(np.float32)).to(DEV)
    W = torch.from_numpy(rs.standard_normal((256, 64)).astype(np.float32)).to(DEV)
    assert run(A, 0, W)[2] == 0
    A2 = A.clone(); A2[17, 3] = 5000.0
    assert run(A2, 0, W)[2] == 1                       # A split in the loop
    W2 = W.clone(); W2[9, 9] = float("inf")
    assert run(A, 0, W2)[2] == 1                       # weight pre-pass
    assert run(A * 300, 0, W * 300, want_f32=False)[2] == 1        # epilogue split of an out-of-range result
    assert run(A * 300, 0, W * 300, want_hr8=False)[2] == 0        # ... which an f32-only output does not care about
    tiny = run(A, 0, W * 1e-6, want_hr8=False)                      # every weight matrix gets its own exponent: tiny weights keep f32-level accuracy
    want = A.double() @ (W.double() * 1e-6).t()
    assert tiny[2] == 0 and bool(((tiny[0].double() - want).abs() <= 2e-6 * want.abs().max()).all())


@pytest.mark.parametrize("M,N,K,act,use_res,want_f32", [(300, 512, 512, 1, False, False), (1000, 256, 2048, 1, False, True), (129, 64, 96, 0, True, True),
                                                          (5000, 512, 8, 1, False, False), (777, 1024, 512, 1, False, False)])
def test_default_gemm_writes_the_split_form_for_the_next_gemm(M, N, K, act, use_res, want_f32):
    """bigc_linear_hr8out_f32: the split-operand GEMM with f32 input (linear_split_kernel; K = 8: linear_k8_kernel) writing its
    result as hr8: decoded, it equals the f32 result of bigc_linear_f32 on the same inputs within the representation error, and
    its f32 twin (when asked for) is bit-identical to it"""
    lib = _lib.load()
    rs = np.random.RandomState(M + N + K)
    A = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(DEV)
    R = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32)).to(DEV) if use_res else None
    want = torch.empty((M, N), device=DEV)
    _lib.check(lib.bigc_linear_f32(_lib.current_stream(), A.data_ptr(), K, None, 1, K, W.data_ptr(), K, b.data_ptr(), _lib.ptr(R), N, want.data_ptr(), N,
                                   M, N, K, act))
    C = torch.full((M, N), float("nan"), device=DEV) if want_f32 else None
    Cs = torch.zeros((M, 4 * N), dtype=torch.uint8, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(lib.bigc_linear_hr8out_f32(_lib.current_stream(), A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), _lib.ptr(R), N, _lib.ptr(C), N,
                                          Cs.data_ptr(), 4 * N, M, N, K, act, flag.data_ptr()))
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    w64 = want.double().cpu().numpy()
    assert (np.abs(hr8_decode(Cs, M, N) - w64) <= repr_tol(w64)).all()
    if want_f32:
        assert torch.equal(C, want)
    # ... and it feeds bigc_gemm_ps directly
    W2 = torch.from_numpy((rs.standard_normal((64, N)) / np.sqrt(N)).astype(np.float32)).to(DEV)
    C2, _, f2 = run(Cs, 1, W2, want_hr8=False, lda_bytes=4 * N)
    assert f2 == 0
    ref = want.double() @ W2.double().t()
    assert float((C2.double() - ref).abs().max() / ref.abs().max()) < 2e-6


def test_default_gemm_split_form_output_raises_the_range_flag():
    lib = _lib.load()
    A = torch.ones((128, 64), device=DEV)
    W = torch.full((16, 64), 100.0, device=DEV)           # outputs 6400 * 8 = 51200 >= 32768 in the scaled form
    Cs = torch.zeros((128, 64), dtype=torch.uint8, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(lib.bigc_linear_hr8out_f32(_lib.current_stream(), A.data_ptr(), 64, W.data_ptr(), 64, None, None, 0, None, 0, Cs.data_ptr(), 64,
                                          128, 16, 64, 0, flag.data_ptr()))
    torch.cuda.synchronize()
    assert int(flag.item()) == 1


@pytest.mark.parametrize("M,K,with_res,with_add,act", [(64 * 5, 512, True, False, 0), (49152 // 8 + 19, 512, True, True, 0), (700, 1024, True, False, 0),
                                                       (100, 64, False, False, 1)])
def test_gemm_ps_with_layernorm_epilogue(M, K, with_res, with_add, act):
    """bigc_gemm_ps_ln: LayerNorm over the 512 output features taken in the GEMM's epilogue (the three norms of a decoder layer,
    models/model_0v10.py:188,211,223) against float64: f32 output (with the optional periodic table added, `+ pos_embedding`), the split
    twin = the normalised rows without the table; ragged M; several tiles per workgroup (the persistent loop) at the larger M"""
    rs = np.random.RandomState(M + K)
    N, period = 512, 192
    A = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(DEV)
    res = torch.from_numpy((2.0 * rs.standard_normal((M, N))).astype(np.float32)).to(DEV) if with_res else None
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, N).astype(np.float32)).to(DEV)
    beta = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(DEV)
    add = torch.from_numpy(rs.standard_normal((period, N)).astype(np.float32)).to(DEV) if with_add else None
    As = to_hr8(A)[0]
    C = torch.full((M, N), float("nan"), device=DEV)
    Cs = torch.zeros((M, 4 * N), dtype=torch.uint8, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(_lib.load().bigc_gemm_ps_ln(_lib.current_stream(), As.data_ptr(), 4 * K, W.data_ptr(), K, bias.data_ptr(), _lib.ptr(res), N, gamma.data_ptr(),
                                           beta.data_ptr(), 1e-5, _lib.ptr(add), period if with_add else 0, C.data_ptr(), N, Cs.data_ptr(), 4 * N, M, N, K, act,
                                           flag.data_ptr()))
    torch.cuda.synchronize()
    x = ref64(A, W, bias, res, act)
    want = torch.nn.functional.layer_norm(x, (N,), gamma.double(), beta.double(), 1e-5)
    back = torch.from_numpy(hr8_decode(Cs, M, N)).to(DEV)
    got = C.double() - (add.double()[torch.arange(M, device=DEV) % period] if with_add else 0.0)
    # the pre-norm row carries the GEMM's round-off (a few 2^-24 of sum |a||w|), LayerNorm divides by the row's standard deviation
    sd = x.std(1, unbiased=False, keepdim=True)
    tol = (8 * 2.0 ** -24 * budget(A, W).amax(1, keepdim=True) + 2.0 ** -22 * x.abs().amax(1, keepdim=True)) / sd * gamma.double().abs() + 2.0 ** -22 * want.abs() + 1e-6
    assert int(flag.item()) == 0
    assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() / tol).max())
    assert bool(((back - want).abs() <= tol + 2.0 ** -22.9 * want.abs()).all())


def test_second_accumulator_variant_in_its_own_process():
    """BIGC_PS_ACC2=all routes the hr8-operand GEMMs with K >= 1024 (and the gathered form) to the 128-row kernel with a second accumulator
    for the correction terms; the switch is read once per process, so the float64 and gather tests above are run again in a child"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_hip_gemm_ps.py"), "-q", "-m", "gpu", "-x", "-k",
                        "vs_float64 or gather_segments or strided"], env=dict(os.environ, BIGC_PS_ACC2="all"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("M,N,K,two_acc", [(700, 512, 2048, 1), (700, 512, 2048, 0), (700, 512, 2048, 2), (256 * 300 + 13, 512, 128, 0), (256 * 300 + 13, 512, 128, 2),
                                           (256 * 300 + 13, 256, 64, 1), (5, 64, 32, 1), (5, 64, 32, 2)])
def test_gemm_ps_rows_by_address(M, N, K, two_acc):
    """bigc_gemm_ps_rows: the A rows given by ADDRESS (fc_feat2enti.0 reading the callers' per-tracklet tensors in place,
    models/model_0v10.py:438,446): raw f32 rows by DMA, split in LDS.  Rows live in separately allocated blocks of different lengths and
    pitches (the visual columns are the first K of wider rows), in shuffled order; against float64; the split twin = the f32 output
    rounded; one accumulator / a second one for the correction terms / blocked accumulation (accum = 0 / 1 / 2; K % 128 != 0 falls back to
    0); the result does not depend on how the rows are scattered."""
    lib = _lib.load()
    rs = np.random.RandomState(M + N + K + two_acc)
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(DEV)
    blocks, ptrs, rows, left = [], [], [], M
    while left > 0:
        n = int(min(left, rs.randint(1, 400)))
        pitch = K + 4 * int(rs.randint(0, 50))                  # wider rows: the GEMM reads the first K floats of each
        blk = torch.from_numpy(rs.standard_normal((n, pitch)).astype(np.float32)).to(DEV)
        blocks.append(blk)
        ptrs += [blk.data_ptr() + 4 * pitch * i for i in range(n)]
        rows.append(blk[:, :K])
        left -= n
    order = rs.permutation(M)
    A = torch.cat(rows)[torch.from_numpy(order).to(DEV)]
    table = torch.from_numpy(np.asarray(ptrs, dtype=np.uint64)[order].view(np.int64)).to(DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    Cs = torch.zeros((M, 4 * N), dtype=torch.uint8, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(lib.bigc_gemm_ps_rows(_lib.current_stream(), table.data_ptr(), W.data_ptr(), K, bias.data_ptr(), C.data_ptr(), N, Cs.data_ptr(), 4 * N,
                                     M, N, K, 1, two_acc, flag.data_ptr()))
    torch.cuda.synchronize()
    want = ref64(A, W, bias, None, 1)
    tol = 6 * 2.0 ** -24 * budget(A, W) + 2.0 ** -23 * want.abs() + 1e-7
    err = (C.double() - want).abs()
    assert int(flag.item()) == 0
    assert bool((err <= tol).all()), "max err/tol %.2f" % float((err / tol).max())
    back = torch.from_numpy(hr8_decode(Cs, M, N)).to(DEV)
    assert bool(((back - C.double()).abs() <= 2.0 ** -22.9 * C.double().abs() + 2.0 ** -28).all())
    # the same rows as ONE dense matrix through bigc_gemm_ps (f32 operand): bit-identical when the tiling is the same (two_acc = 0 picks it from M alone)
    if not two_acc:
        C2, _, _ = run(A.contiguous(), 0, W, bias, None, 1, want_hr8=False)
        assert torch.equal(C2, C)
    # a NaN / an out-of-range value among the rows raises the flag
    blocks[0][0, 1] = float("nan")
    _lib.check(lib.bigc_gemm_ps_rows(_lib.current_stream(), table.data_ptr(), W.data_ptr(), K, bias.data_ptr(), C.data_ptr(), N, None, 0, M, N, K, 1, two_acc, flag.data_ptr()))
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    blocks[0][0, 1] = 4096.0
    flag.zero_()
    _lib.check(lib.bigc_gemm_ps_rows(_lib.current_stream(), table.data_ptr(), W.data_ptr(), K, bias.data_ptr(), C.data_ptr(), N, None, 0, M, N, K, 1, two_acc, flag.data_ptr()))
    torch.cuda.synchronize()
    assert int(flag.item()) == 1


def test_gemm_ps_accumulation_modes_round_less():
    """PsGemm.accum: rms rounding error against float64 on the same operands, in units of 2^-24 rms(C) at K = 2048: one accumulator >
    second accumulator for the correction terms > blocked accumulation (the model's default, csrc/model.hip ps_accum), the last below
    torch's CPU GEMM"""
    lib = _lib.load()
    rs = np.random.RandomState(3)
    M, N, K = 2048, 512, 2048
    A = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rs.standard_normal((N, K)) * (2.0 / (N + K)) ** 0.5).astype(np.float32)).to(DEV)
    ref = A.double() @ W.double().t()
    unit = float(ref.pow(2).mean().sqrt()) * 2.0 ** -24
    table = torch.from_numpy((A.data_ptr() + 4 * K * np.arange(M, dtype=np.int64))).to(DEV)
    out = {}
    for two in (0, 1, 2):
        C = torch.empty((M, N), device=DEV)
        _lib.check(lib.bigc_gemm_ps_rows(_lib.current_stream(), table.data_ptr(), W.data_ptr(), K, None, C.data_ptr(), N, None, 0, M, N, K, 0, two, None))
        torch.cuda.synchronize()
        out[two] = float((C.double() - ref).pow(2).mean().sqrt()) / unit
    # measured: 8.5 / 5.0 / 2.9 units (torch's CPU GEMM on the same operands: 4.3)
    assert out[2] < out[1] < out[0] and out[1] < 7.0 and out[2] < 4.0, out
