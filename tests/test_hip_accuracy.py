"""Accuracy and range behaviour of the default GEMM (csrc/linear_split.hip: f32 operands as two f16 halves, three f16 MFMAs per
product, in-kernel fallback to the exact f32 MFMA) and of the whole eval forward, as assertions (they used to be scripts:
benchmarks/linear_accuracy.py, tests/accuracy_vs_fp64.py)."""
import copy
import os
import tempfile

import numpy as np
import pytest
import torch

from bigc_hip import _lib, configs
from bigc_hip.packed import PackedBatch
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_proposal

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run_linear(A, W, b=None, res=None, act=0):
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    C = torch.full((M, N), float("nan"), device=A.device)
    _lib.check(lib.bigc_linear_f32(_lib.current_stream(), A.data_ptr(), K, None, 1, K, W.data_ptr(), K, _lib.ptr(b), _lib.ptr(res), N, C.data_ptr(), N,
                                   M, N, K, act))
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize("name,M,N,K,sa,sw,bound", [("normal x 0.02", 2048, 512, 2048, 1.0, 0.02, 4.0), ("relu x xavier", 2048, 512, 512, 3.0, 0.06, 4.0),
                                                    ("large", 1024, 512, 512, 3e3, 5.0, 4.0), ("tiny", 1024, 512, 512, 1e-6, 1e-4, 128.0),
                                                    ("wide range", 1024, 133, 3160, 1.0, 0.02, 128.0)])
def test_split_gemm_error_is_at_the_f32_level(name, M, N, K, sa, sw, bound):
    """error of the split-operand GEMM against float64, in units of the f32 round-off scale sum |a||w| (2^-24 = one f32
    round-off): within 4 units -- torch's own f32 GEMM on the CPU measures 1 on these shapes, a sequential fmaf chain K/8 --
    for operands of ordinary size.  Documented limits, asserted as such: operands BELOW the f16 normal range (|x| < 6e-5: the
    "tiny" case, where both halves of the split are f16 subnormals) only keep an absolute error of 1.5e-11 per element, measured 53
    units; tiles with an operand >= 32768 ("wide range") take the exact f32 MFMA, a K-long fmaf chain, measured 69 units at K = 3160."""
    g = torch.Generator(device=DEV).manual_seed(1)
    A = torch.randn(M, K, device=DEV, generator=g) * sa
    if name == "relu x xavier":
        A = A.clamp_min(0)
    if name == "wide range":
        A = A * torch.exp(4 * torch.randn(M, K, device=DEV, generator=g))
    W = torch.randn(N, K, device=DEV, generator=g) * sw
    b = torch.randn(N, device=DEV, generator=g) * sa * sw
    C = run_linear(A, W, b)
    ref = A.double() @ W.double().t() + b.double()
    scale = A.double().abs() @ W.double().abs().t() + b.double().abs()
    err = float(((C.double() - ref).abs() / scale).max())
    assert err <= bound * 2.0 ** -24, (name, err * 2 ** 24)


def test_gemm_range_fallback_large_nan_inf():
    """operands the f16 split cannot hold (|x| >= 32768, NaN, Inf), in A and in W: the tile is redone on the exact f32 MFMA inside
    the kernel, so the result equals an f32 GEMM's -- finite where it is, NaN / Inf rows and columns where torch's are"""
    g = torch.Generator(device=DEV).manual_seed(2)
    M, N, K = 700, 300, 256
    A0 = torch.randn(M, K, device=DEV, generator=g)
    W0 = torch.randn(N, K, device=DEV, generator=g) * 0.1
    b = torch.randn(N, device=DEV, generator=g)

    def check(A, W):
        C = run_linear(A, W, b)
        ref = (A.double() @ W.double().t() + b.double())
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(C), fin)
        assert torch.equal(torch.isnan(C), torch.isnan(ref))
        scale = (A.double().abs() @ W.double().abs().t() + b.double().abs())
        e = ((C.double() - ref).abs() / scale)[fin]
        assert float(e.max()) <= 64 * 2.0 ** -24, float(e.max()) * 2 ** 24          # the fallback is an fmaf chain of K = 256 terms
        inf = torch.isinf(ref)
        assert torch.equal(torch.sign(C[inf]).double(), torch.sign(ref[inf]))

    A = A0.clone(); A[7, 3] = 1e6; A[100, 200] = -1e9; A[650, 0] = 40000.0
    check(A, W0)                                            # large finite activations
    W = W0.clone(); W[5, 9] = 70000.0; W[299, 255] = -3e7
    check(A0, W)                                            # large finite weights
    A = A0.clone(); A[5, 5] = float("inf"); A[300, 1] = float("nan"); A[301, 2] = float("-inf")
    check(A, W0)
    W = W0.clone(); W[17, 100] = float("nan"); W[18, 3] = float("inf")
    check(A0, W)
    A = A0.clone(); A[0, 0] = 32768.0                       # exactly at the limit
    check(A, W0)
    check(A0 * 32767.0 / float(A0.abs().max()), W0)        # just inside: stays on the split path, must still be accurate


def test_eval_logits_accuracy_vs_float64():
    """benchmark model (exp2 dims, sharp attention), 4 videos: distance of the HIP logits to a float64 evaluation of the oracle is
    at the level of the f32 oracle's own (the reference's arithmetic), in both GEMM modes; attention arg-max equal to float64's"""
    from models import BIG_C_vidvrd
    from oracle import bigc_oracle as O
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    d = tempfile.mkdtemp()
    np.save(d + "/b.npy", sd["bias_matrix"].numpy()); np.save(d + "/e.npy", sd["EntiNameEmb"].numpy())
    model = BIG_C_vidvrd(dict(cfg, bias_matrix_path=d + "/b.npy", EntiNameEmb_path=d + "/e.npy"), is_train=False)
    model.load_state_dict(sd)
    model = model.to(DEV)
    lib = _lib.load()
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    sd64 = {k: v.double() for k, v in sd.items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    worst = {"split": 0.0, "exact": 0.0, "cpu32": 0.0}
    for v in (0, 2, 3, 5):
        p = synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
        with torch.no_grad():
            o32 = O.encode2decode(sd, cfg, p)
            p64 = copy.deepcopy(p)
            p64.bboxes_list = [b.double() for b in p64.bboxes_list]
            p64.features_list = [f.double() for f in p64.features_list]
            p64.scores = p64.scores.double()
            o64 = O.encode2decode(sd64, cfg, p64)
        ref = o64["pred_logits"]
        scale = float(ref.abs().max())
        packed = PackedBatch([copy.deepcopy(p).to(DEV)], DEV)
        for name, mode in (("split", 0), ("exact", 1)):
            prev = lib.bigc_linear_set_exact(mode)
            try:
                out = model.forward_packed(packed, 10)
            finally:
                lib.bigc_linear_set_exact(prev)
            worst[name] = max(worst[name], float((out["pred_logits"].double().cpu() - ref).abs().max()) / scale)
            if name == "split":
                assert torch.equal(out["soid"][0].cpu().long(), torch.argmax(o64["att"], dim=-1))
        worst["cpu32"] = max(worst["cpu32"], float((o32["pred_logits"].double() - ref).abs().max()) / scale)
    # measured: split 5.4e-5, f32 oracle 1.9e-5, exact-f32-MFMA mode 1.2e-4 (its K-long fmaf chains)
    assert worst["split"] <= 1e-4 and worst["exact"] <= 5e-4, worst
