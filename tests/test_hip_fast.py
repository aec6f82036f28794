"""GPU tests of the opt-in reduced-precision GEMM mode (bigc_linear_set_fast / BIGC_FAST=1; csrc/linear_split.hip and
csrc/gemm_ps.hip with FAST = true): every operand rounded to ONE f16, one f16 MFMA per product, f32 accumulation.  It is the
"bf16 fast mode" of SURVEY.md section 7 step 10: not held to the 1e-4 logit tolerance, gated on Recall@50 parity (section 8d:
|R@50(fast) - R@50(parity)| <= 1e-3 on the synthetic-GT set)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import cases  # noqa: E402
from bigc_hip import _lib  # noqa: E402
from bigc_hip import evaluate as E  # noqa: E402
from bigc_hip.packed import PackedBatch  # noqa: E402
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_gt_graph  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class fast_mode:
    def __enter__(self):
        self.lib = _lib.load()
        self.prev = self.lib.bigc_linear_set_fast(1)
        return self.lib

    def __exit__(self, *a):
        self.lib.bigc_linear_set_fast(self.prev)


def f16_round(x):
    return x.half().float()


def test_mode_switch_returns_the_previous_mode_and_yields_to_exact_mode():
    lib = _lib.load()
    prev = lib.bigc_linear_set_fast(1)
    try:
        assert lib.bigc_linear_get_fast() == 1 and lib.bigc_linear_set_fast(1) == 1
        ex = lib.bigc_linear_set_exact(1)
        try:
            assert lib.bigc_linear_get_fast() == 0            # the exact mode wins
        finally:
            lib.bigc_linear_set_exact(ex)
    finally:
        lib.bigc_linear_set_fast(prev)
    assert lib.bigc_linear_get_fast() == prev


@pytest.mark.parametrize("M,N,K,res", [(1000, 512, 512, False), (700, 133, 300, True), (4096, 1024, 2048, False)])
def test_fast_linear_equals_the_f16_rounded_product(M, N, K, res):
    """bigc_linear_f32 in fast mode = (f16(A) f16(W)^T accumulated in f32) + bias + residual: products of f16 pairs are exact in
    f32, so the only difference to a float64 evaluation of the ROUNDED operands is the f32 accumulation (a few 2^-24 of sum |a||w|)"""
    g = torch.Generator(device=DEV).manual_seed(3)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * 0.05
    b = torch.randn(N, device=DEV, generator=g)
    R = torch.randn(M, N, device=DEV, generator=g) if res else None
    C = torch.full((M, N), float("nan"), device=DEV)
    with fast_mode() as lib:
        _lib.check(lib.bigc_linear_f32(_lib.current_stream(), A.data_ptr(), K, None, 1, K, W.data_ptr(), K, b.data_ptr(), _lib.ptr(R), N, C.data_ptr(), N,
                                       M, N, K, 1))
        torch.cuda.synchronize()
    Ar, Wr = f16_round(A).double(), f16_round(W).double()
    ref = torch.relu(Ar @ Wr.t() + b.double())
    if res:
        ref = ref + R.double()
    scale = Ar.abs() @ Wr.abs().t() + b.double().abs()
    assert float(((C.double() - ref).abs() / scale).max()) < 8 * 2.0 ** -24
    # ... and it is NOT the full-precision product: the rounding of the operands shows (a few 1e-4 of the scale)
    full = torch.relu(A.double() @ W.double().t() + b.double()) + (R.double() if res else 0)
    assert float(((C.double() - full).abs() / scale).max()) > 2.0 ** -16


@pytest.mark.parametrize("M,N,K,res", [(1000, 512, 512, False), (700, 133, 300, True), (4096, 1024, 2048, False)])
def test_bf16_linear_equals_the_bf16_rounded_product(M, N, K, res):
    """mode 2 (BIGC_FAST=bf16 / bigc_linear_set_fast(2)), the literal bf16 arithmetic of BASELINE configs[1]: bigc_linear_f32 =
    (bf16(A) bf16(W)^T accumulated in f32) + bias + residual.  Products of bf16 pairs are exact in f32; the weight's bf16 is rounded
    from its cached split form (22 bits of it), so an element sitting within 2^-22 of a bf16 rounding boundary may round the other
    way: the bound allows for one such element per dot product on top of the f32 accumulation."""
    g = torch.Generator(device=DEV).manual_seed(5)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * 0.05
    b = torch.randn(N, device=DEV, generator=g)
    R = torch.randn(M, N, device=DEV, generator=g) if res else None
    C = torch.full((M, N), float("nan"), device=DEV)
    lib = _lib.load()
    prev = lib.bigc_linear_set_fast(2)
    try:
        assert lib.bigc_linear_get_fast() == 2
        _lib.check(lib.bigc_linear_f32(_lib.current_stream(), A.data_ptr(), K, None, 1, K, W.data_ptr(), K, b.data_ptr(), _lib.ptr(R), N, C.data_ptr(), N,
                                       M, N, K, 1))
        torch.cuda.synchronize()
    finally:
        lib.bigc_linear_set_fast(prev)
    Ar, Wr = A.bfloat16().double(), W.bfloat16().double()
    ref = torch.relu(Ar @ Wr.t() + b.double())
    if res:
        ref = ref + R.double()
    scale = Ar.abs() @ Wr.abs().t() + b.double().abs()
    err = float(((C.double() - ref).abs() / scale).max())
    assert err < 8 * 2.0 ** -24 + 2.0 ** -8 / K * 4, err          # f32 accumulation + (rarely) one weight element rounded the other way
    # it is the bf16 product, not the f16-rounded one and not the full-precision one
    f16ref = torch.relu(A.half().double() @ W.half().double().t() + b.double()) + (R.double() if res else 0)
    assert float(((C.double() - f16ref).abs() / scale).max()) > 2.0 ** -13
    assert not torch.isnan(C).any()


def test_bf16_mode_of_a_model_handle():
    """bigc_set_mode(handle, fast = 2): the model forward in the bf16 arithmetic (no GEMM -> GEMM chains through the f16 split form).
    It runs, differs from the parity mode by bf16-sized amounts, and leaves no trace once the handle is back in the default mode
    (its Recall@50 gate is bench.py's `recall.bf16_gate_passed`, on a trained model)."""
    cfg, variant, sd, props, _topk = cases.make_case("bigc_v_n64")
    from test_hip_bigc import build_model
    model = build_model(cfg, variant, sd)
    dev_props = [p.to(DEV) for p in props]
    packed = PackedBatch([p for p in dev_props if p.num_proposals > 0], DEV)
    ref = model.forward_packed(packed, 10)
    torch.cuda.synchronize()
    ref_logits = ref["pred_logits"].clone()
    model.engine().set_mode(fast=2)
    try:
        out = model.forward_packed(packed, 10)
        torch.cuda.synchronize()
        d = float((out["pred_logits"] - ref_logits).abs().max() / ref_logits.abs().max())
    finally:
        model.engine().set_mode(fast=-1)
    assert 1e-5 < d < 10.0, d          # (sharp-attention seeded weights: bf16 flips arg-max entities, whole logit rows change)
    assert torch.isfinite(out["pred_logits"]).all()
    again = model.forward_packed(packed, 10)
    torch.cuda.synchronize()
    assert torch.equal(again["pred_logits"], ref_logits)


def test_fast_gemm_ps_reads_the_heads_only():
    """bigc_gemm_ps on an hr8 operand in fast mode = product of the f16 heads of both operands (activations stored as x * 8,
    weights scaled per matrix by a power of two: the heads are f16(x * 2^e) / 2^e), f32 and hr8 outputs"""
    from test_hip_gemm_ps import hr8_decode, repr_tol, to_hr8
    M, N, K = 777, 512, 512
    g = torch.Generator(device=DEV).manual_seed(5)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * 0.05
    b = torch.randn(N, device=DEV, generator=g)
    As, _ = to_hr8(A)
    C = torch.full((M, N), float("nan"), device=DEV)
    Cs = torch.zeros((M, 4 * N), dtype=torch.uint8, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    with fast_mode() as lib:
        _lib.check(lib.bigc_gemm_ps(_lib.current_stream(), As.data_ptr(), 4 * K, 1, None, 1, W.data_ptr(), K, b.data_ptr(), None, 0, C.data_ptr(), N,
                                    Cs.data_ptr(), 4 * N, M, N, K, 0, flag.data_ptr()))
        torch.cuda.synchronize()
    assert int(flag.item()) == 0
    ew = 14 - int(np.floor(np.log2(float(W.abs().max())))) - 1         # max |W| 2^ew in [2^13, 2^14)
    Ar = (f16_round(A * 8.0) / 8.0).double()
    Wr = (f16_round(W * 2.0 ** ew) / 2.0 ** ew).double()
    ref = Ar @ Wr.t() + b.double()
    scale = Ar.abs() @ Wr.abs().t() + b.double().abs()
    assert float(((C.double() - ref).abs() / scale).max()) < 8 * 2.0 ** -24
    c64 = C.double().cpu().numpy()
    assert (np.abs(hr8_decode(Cs, M, N) - c64) <= repr_tol(c64)).all()                            # the hr8 twin of the f32 output


def _model(cfg, sd):
    from models import BIG_C_vidvrd
    d = tempfile.mkdtemp()
    mcfg = dict(cfg)
    np.save(d + "/b.npy", sd["bias_matrix"].numpy())
    np.save(d + "/e.npy", sd["EntiNameEmb"].numpy())
    mcfg["bias_matrix_path"], mcfg["EntiNameEmb_path"] = d + "/b.npy", d + "/e.npy"
    model = BIG_C_vidvrd(mcfg, is_train=False)
    model.load_state_dict(sd)
    return model.to(DEV)


@pytest.mark.parametrize("cfg_name,n_videos,n_trajs,topk", [("CONFIG_TINY", 12, 14, 3), ("CONFIG_V", 6, 40, 10)])
def test_recall_gate_of_the_fast_mode(cfg_name, n_videos, n_trajs, topk):
    """|R@50(fast) - R@50(parity)| <= 1e-3 on the synthetic-GT set; and the fast mode is a DIFFERENT arithmetic
    (its logits are not those of the parity mode) that stays within a few percent of them"""
    cfg = cases.get_cfg(cfg_name)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 61, att_gain=6.0)
    model = _model(cfg, sd)
    props, gts = [], {}
    for v in range(n_videos):
        p = cases.make_proposal(cfg, "vidvrd", dict(seed=3000 + v, n_trajs=n_trajs, len_range=(8, 40), start_range=(0, 20)))
        p.video_name = "synth_%d" % v
        g = synthetic_gt_graph(p, 4000 + v, n_gt_trajs=6, n_preds=8, num_pred_cats=cfg["num_pred_cats"], jitter=0.0)
        props.append(p)
        gts[p.video_name] = E.gt_to_eval_format(g)
    packed = PackedBatch([p.to(DEV) for p in props], DEV)

    def run():
        out = model.forward_packed(packed, topk)
        res = model.unpack_results(len(props), list(range(len(props))), out)
        pred = {}
        for p, r in zip(props, res):
            pred.update(E.to_eval_format(p, r))
        return out["pred_logits"].clone(), out["soid"].clone(), pred

    lg_p, so_p, pred_p = run()
    with fast_mode():
        lg_f, so_f, pred_f = run()
    _, r_p, _ = E.evaluate(gts, pred_p)
    _, r_f, _ = E.evaluate(gts, pred_f)
    assert abs(r_p[50] - r_f[50]) <= 1e-3 and abs(r_p[100] - r_f[100]) <= 1e-3, (r_p, r_f)
    same = (so_p == so_f).all(1).reshape(-1)
    assert float(same.float().mean()) > 0.9                                      # the attention arg-max survives for most queries
    rel = float(((lg_p - lg_f).abs().max(1).values * same).max() / lg_p.abs().max())
    assert 1e-6 < rel < 0.1, rel


def test_arithmetic_mode_is_per_handle_and_per_thread():
    """bigc_set_mode: the arithmetic mode travels with the model handle ("re-entrant per handle"; the reference's
    DataParallel runs one replica per thread, utils/DataParallel.py:68).  Three models with the same weights in one process -- default,
    exact f32 MFMA, reduced precision: each gives, bit for bit, what the whole process gives under the corresponding process-wide
    switch, the process-wide switches stay untouched, and running the three from three threads at once changes nothing."""
    import threading
    from test_hip_bigc import build_model
    from bigc_hip.synthetic import synthetic_proposal
    lib = _lib.load()
    cfg = cases.get_cfg("CONFIG_TINY")
    cfg = dict(cfg, dim_enti=64, dim_pred=64, dim_att=64, dim_ffn=64, dim_feat=128, dim_i3d=32, n_att_head=4, num_querys=24)     # wide enough for the production kernels
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 91, att_gain=3.0)
    props = [synthetic_proposal(4400 + v, n_trajs=20, feat_width=160, num_enti_cats=cfg["num_enti_cats"], len_range=(6, 40), start_range=(0, 20)).to(DEV)
             for v in range(6)]
    packed = PackedBatch(props, DEV)
    keys = ("pred_logits", "soid", "n_out", "rows")

    def run(model):
        out = model.forward_packed(packed, 3)
        n = model.counts(out).tolist()
        res = {k: out[k].clone() for k in ("pred_logits", "soid", "n_out")}
        # the triplet arrays are padded to a fixed capacity per video: only the first n_out[v] rows are defined
        res["rows"] = torch.cat([torch.cat([out["quint"][v, :n[v]].float(), out["scores"][v, :n[v]], out["dura"][v, :n[v]].float()], 1) for v in range(len(n))])
        return res

    # references under the process-wide switches
    base = build_model(cfg, "vidvrd", sd)
    want = {"default": run(base)}
    prev = lib.bigc_linear_set_exact(1)
    try:
        want["exact"] = run(base)
    finally:
        lib.bigc_linear_set_exact(prev)
    prev = lib.bigc_linear_set_fast(1)
    try:
        want["fast"] = run(base)
    finally:
        lib.bigc_linear_set_fast(prev)
    assert not torch.equal(want["default"]["pred_logits"], want["exact"]["pred_logits"])
    assert not torch.equal(want["default"]["pred_logits"], want["fast"]["pred_logits"])
    # one handle per mode
    models = {"default": build_model(cfg, "vidvrd", sd), "exact": build_model(cfg, "vidvrd", sd), "fast": build_model(cfg, "vidvrd", sd)}
    models["exact"].engine().set_mode(exact=1)
    models["fast"].engine().set_mode(fast=1)
    for _ in range(2):
        for name in ("fast", "default", "exact"):
            got = run(models[name])
            assert all(torch.equal(got[k], want[name][k]) for k in keys), name
    assert lib.bigc_linear_get_exact() == 0 and lib.bigc_linear_get_fast() == 0          # the process-wide switches were never touched
    # ... and from three threads at once (each call opens its own thread-local scope)
    results, errors = {}, []

    def worker(name):
        try:
            torch.cuda.set_device(0)
            for _ in range(5):
                results[name] = run(models[name])
        except Exception as e:                         # noqa: BLE001
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=worker, args=(n,)) for n in models]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for name in models:
        assert all(torch.equal(results[name][k], want[name][k]) for k in keys), name
