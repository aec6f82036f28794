"""GPU parity of the grounding scorer (HIP path through the C ABI) against the reference goldens and
the CPU oracle: network outputs within 1e-4 relative, pooled spans / bin probabilities within 1e-5,
bin masks exact."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import cases
from bigc_hip import _lib
from oracle import grd_oracle as G

DEV = "cuda:0"


def S():
    return _lib.current_stream()


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def build_model(cfg, sd):
    from models import DEBUG
    d = tempfile.mkdtemp()
    cfg = dict(cfg)
    np.save(os.path.join(d, "e.npy"), sd["EntiNameEmb"].numpy())
    np.save(os.path.join(d, "p.npy"), sd["PredNameEmb"].numpy())
    cfg["EntiNameEmb_path"], cfg["PredNameEmb_path"] = os.path.join(d, "e.npy"), os.path.join(d, "p.npy")
    m = DEBUG(cfg, is_train=False)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV)


@pytest.mark.parametrize("n_seq,T,d,ks", [(1, 9, 64, 7), (5, 3, 128, 3), (7, 40, 128, 7), (3, 1, 16, 3), (2, 2, 32, 7)])
def test_dwconv(n_seq, T, d, ks):
    lib = _lib.load()
    g = torch.Generator().manual_seed(n_seq * 100 + T)
    x = torch.randn(n_seq, T, d, generator=g)
    w = torch.randn(d, 1, ks, generator=g)
    b = torch.randn(d, generator=g)
    want = F.conv1d(x.transpose(1, 2), w, b, padding=ks // 2, groups=d).transpose(1, 2)
    xd, wd, bd = x.to(DEV).contiguous(), w.to(DEV).contiguous(), b.to(DEV)
    y = torch.full((n_seq * T, d), float("nan"), device=DEV)
    _lib.check(lib.bigc_dwconv_f32(S(), xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), n_seq, T, d, ks))
    torch.cuda.synchronize()
    assert rel(y.cpu().reshape(n_seq, T, d), want) < 2e-6


@pytest.mark.parametrize("n,T,d", [(5, 9, 64), (1, 300, 128), (17, 2, 128), (3, 1000, 128)])
def test_coattention(n, T, d):
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + T)
    video = torch.randn(T, d, generator=g)
    proj = torch.randn(T, d, generator=g) * 0.3
    query = torch.randn(n, 3, d, generator=g)
    sim = torch.matmul(proj[None].expand(n, -1, -1), query.transpose(1, 2))            # (n,T,3)
    sr, sc = torch.softmax(sim, 2), torch.softmax(sim, 1)
    v = video[None].expand(n, -1, -1)
    a = sr @ query
    b = (sr @ sc.transpose(1, 2)) @ v
    want = torch.cat([v, a, a * v, b * v], -1)
    vd, pd, qd = video.to(DEV), proj.to(DEV), query.to(DEV).contiguous()
    out = torch.full((n * T, 4 * d), float("nan"), device=DEV)
    _lib.check(lib.grd_coattention_f32(S(), vd.data_ptr(), pd.data_ptr(), qd.data_ptr(), out.data_ptr(), n, T, d))
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    assert rel(out.cpu().reshape(n, T, 4 * d), want) < 1e-5


def run_postprocess(regrs, conf, cls, info, k, th):
    lib = _lib.load()
    n, T = conf.shape[0], conf.shape[1]
    rd, cd, ld, idv = [t.to(DEV).contiguous() for t in (regrs, conf, cls, info)]
    clip = torch.linspace(0, 1, T).to(DEV)
    pooled = torch.empty(n, k + 1, 2, device=DEV)
    probs = torch.empty(n, k + 1, device=DEV)
    mask = torch.empty(n, k + 1, dtype=torch.uint8, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(lib.grd_postprocess(S(), rd.data_ptr(), cd.data_ptr(), ld.data_ptr(), clip.data_ptr(), idv.data_ptr(), n, T, k,
                                   th["score_th"], th["tiou_th"], th["bins_th"], th["nms_th"], pooled.data_ptr(), probs.data_ptr(),
                                   mask.data_ptr(), status.data_ptr()))
    torch.cuda.synchronize()
    return pooled.cpu(), probs.cpu(), mask.cpu().bool(), int(status.item())


@pytest.mark.parametrize("name", list(cases.GRD_CASES))
def test_postprocess_on_reference_network_outputs(name):
    """a14-a16 alone, fed with the REFERENCE's regrs/conf/cls: spans to 1e-6, masks exact."""
    g = np.load(os.path.join(cases.GOLDEN_DIR, name + ".npz"))
    cfg, sd, videos = cases.make_grd_case(name)
    for v, (feat, data) in enumerate(videos):
        pre = "v%d_" % v
        if pre + "regrs" not in g.files:             # outputs-only fixture
            continue
        info = data[1].float() / data[2]
        pooled, probs, mask, status = run_postprocess(torch.from_numpy(g[pre + "regrs"]), torch.from_numpy(g[pre + "conf"]),
                                                      torch.from_numpy(g[pre + "cls"]), info, cfg["num_bins"], cases.GRD_TH)
        assert status == 0
        assert np.abs(pooled.numpy() - g[pre + "pooled_se"]).max() < 2e-6
        assert np.abs(probs.numpy() - g[pre + "bins_probs"]).max() < 2e-6
        assert np.array_equal(mask.numpy(), g[pre + "bins_mask"])


def test_postprocess_thresholds_and_nms_vs_oracle():
    """random heads with other thresholds (heavy NMS suppression, weak bins, disjoint spans)."""
    rs = np.random.RandomState(3)
    n, T, k = 40, 25, 10
    regrs = torch.from_numpy(rs.uniform(0.01, 0.4, (n, T, 2 * k)).astype(np.float32))
    conf = torch.from_numpy((3 * rs.standard_normal((n, T, k))).astype(np.float32))
    cls = torch.from_numpy((3 * rs.standard_normal((n, T, k))).astype(np.float32))
    conf[:5] -= 8.0                                           # weak queries: no real bin above bins_th
    s = rs.uniform(0, 0.6, n)
    info = torch.from_numpy(np.stack([s, s + rs.uniform(0.01, 0.4, n)], 1).astype(np.float32))
    for th in (dict(score_th=0.5, tiou_th=0.3, bins_th=0.3, nms_th=0.4), dict(score_th=0.9, tiou_th=0.5, bins_th=0.2, nms_th=0.8)):
        pooled, probs, mask, status = run_postprocess(regrs, conf, cls, info, k, th)
        scores = conf.sigmoid() * cls.sigmoid()
        want_pool = G.temporal_pooling(regrs, scores, k, th["score_th"], th["tiou_th"])
        assert np.abs(pooled[:, :k].numpy() - _clip_to(want_pool, info).numpy()).max() < 2e-6
        # full post-processing through the oracle's own function with a stub network
        wp, wprob, wmask = _oracle_post(regrs, conf, cls, info, k, th)
        assert status == 0
        assert np.abs(pooled.numpy() - wp.numpy()).max() < 2e-6
        assert np.abs(probs.numpy() - wprob.numpy()).max() < 2e-6
        assert np.array_equal(mask.numpy(), wmask.numpy())


def _clip_to(pooled, info):
    s = torch.maximum(info[:, None, 0], pooled[:, :, 0])
    e = torch.minimum(info[:, None, 1], pooled[:, :, 1])
    ov = s <= e
    return torch.where(ov[:, :, None], torch.stack([s, e], -1), info[:, None, :].expand(-1, pooled.shape[1], -1))


def _oracle_post(regrs, conf, cls, info, k, th):
    orig = G.forward_propagation
    G.forward_propagation = lambda sd, vf, we, idura: (regrs, conf, cls)
    try:
        return G.forward_test_single(None, None, None, info, k, th["score_th"], th["tiou_th"], th["bins_th"], th["nms_th"])
    finally:
        G.forward_propagation = orig


@pytest.mark.parametrize("name", list(cases.GRD_CASES))
def test_grounding_vs_reference_golden(name):
    g = np.load(os.path.join(cases.GOLDEN_DIR, name + ".npz"))
    cfg, sd, videos = cases.make_grd_case(name)
    model = build_model(cfg, sd)
    k = cfg["num_bins"]
    for v, (feat, data) in enumerate(videos):
        pre = "v%d_" % v
        quint, dura, video_len = data
        n, T = quint.shape[0], feat.shape[0]
        raw = model.forward_raw(feat.to(DEV), quint, dura, video_len, keep_intermediates=True, **cases.GRD_TH)
        torch.cuda.synchronize()
        if pre + "regrs" not in g.files:             # outputs-only fixture: the final spans / probabilities / masks of the reference
            pooled, probs, mask = model([feat.to(DEV)], [data], with_gt_data=False, **cases.GRD_TH)
            assert np.abs(pooled.cpu().numpy() - g[pre + "pooled_se"]).max() < 2e-5
            assert np.abs(probs.cpu().numpy() - g[pre + "bins_probs"]).max() < 2e-5
            assert np.array_equal(mask.cpu().numpy(), g[pre + "bins_mask"])
            continue
        if pre + "video_enc" in g.files:
            assert rel(raw["video_enc"].cpu().numpy(), g[pre + "video_enc"]) < 2e-5
            assert rel(raw["query_enc"].cpu().numpy().reshape(n, 3, -1), g[pre + "query_enc"]) < 2e-5
            assert rel(raw["combined_enc"].cpu().numpy().reshape(n, T, -1), g[pre + "combined_enc"]) < 1e-4
        assert rel(raw["regrs"].cpu().numpy().reshape(n, T, 2 * k), g[pre + "regrs"]) < 1e-4
        assert rel(raw["conf"].cpu().numpy().reshape(n, T, k), g[pre + "conf"]) < 1e-4
        assert rel(raw["cls"].cpu().numpy().reshape(n, T, k), g[pre + "cls"]) < 1e-4
        pooled, probs, mask = model([feat.to(DEV)], [data], with_gt_data=False, **cases.GRD_TH)
        assert mask.dtype == torch.bool and pooled.shape == (n, k + 1, 2)
        assert np.abs(pooled.cpu().numpy() - g[pre + "pooled_se"]).max() < 2e-5
        assert np.abs(probs.cpu().numpy() - g[pre + "bins_probs"]).max() < 2e-5
        assert np.array_equal(mask.cpu().numpy(), g[pre + "bins_mask"])


def test_grounding_empty_and_all_zero_scores():
    cfg, sd, videos = cases.make_grd_case("grd_tiny")
    model = build_model(cfg, sd)
    feat, (quint, dura, vl) = videos[0]
    assert model([feat.to(DEV)], [(quint[:0], dura[:0], vl)], with_gt_data=False) == (None, None)
    # all scores exactly zero: the reference dies with an IndexError in temporal_pooling (:726)
    n, T, k = 2, 5, 3
    regrs = torch.full((n, T, 2 * k), 0.1)
    neg = torch.full((n, T, k), -200.0)
    info = torch.tensor([[0.1, 0.5], [0.2, 0.9]])
    _, _, _, status = run_postprocess(regrs, neg, neg, info, k, cases.GRD_TH)
    assert status == 1


def test_model_forward_raises_where_the_reference_raises_on_all_zero_scores():
    """``temporal_pooling`` takes ``torch.min`` of an EMPTY selection when a (query, bin) column of scores is all zero -- the
    reference dies with an IndexError (models/grd_model_v5.py:726).  The convention here: the kernel raises the ``status`` word of the
    video (nothing else of its output is defined), ``DEBUG.forward`` turns it into the same IndexError, the batched entry point hands the
    status out per video (tools/eval_vidor.py skips such a video), and the CPU oracle asserts at the same place.  Forced through the whole
    model: head biases of -1e4 make sigmoid(conf) and sigmoid(cls) exactly 0."""
    from oracle import grd_oracle as G
    cfg, sd, videos = cases.make_grd_case("grd_tiny")
    sd = {k: v.clone() for k, v in sd.items()}
    last = sorted(k for k in sd if k.startswith("cls_head.") and k.endswith(".bias"))[-1]
    sd[last] = torch.full_like(sd[last], -1e4)
    model = build_model(cfg, sd)
    feat, (quint, dura, vl) = videos[0]
    with pytest.raises(IndexError):
        model([feat.to(DEV)], [(quint, dura, vl)], with_gt_data=False, **cases.GRD_TH)
    outs = model.forward_raw_batch([(feat.to(DEV), quint, dura, vl)], **cases.GRD_TH)
    assert int(outs[0]["status"].item()) == 1
    with pytest.raises(AssertionError):
        G.forward(sd, cfg, feat, (quint, dura, vl), **cases.GRD_TH)


def test_grounding_batch_of_videos_vs_reference_golden_and_vs_one_video_calls():
    """grd_forward_batch on the three grd_tiny videos (T = 9 / 33 / 2 clips, n = 5 / 1 / 4 queries) in ONE call: every video's
    result matches the reference golden and is bit-identical to the one-video-per-call forward."""
    g = np.load(os.path.join(cases.GOLDEN_DIR, "grd_tiny.npz"))
    cfg, sd, videos = cases.make_grd_case("grd_tiny")
    model = build_model(cfg, sd)
    jobs = [(feat.to(DEV), quint, dura, vl) for feat, (quint, dura, vl) in videos]
    outs = model.forward_raw_batch(jobs, **cases.GRD_TH)
    torch.cuda.synchronize()
    assert len(outs) == len(videos)
    for v, (o, job) in enumerate(zip(outs, jobs)):
        pre = "v%d_" % v
        assert int(o["status"].item()) == 0
        assert np.abs(o["pooled_se"].cpu().numpy() - g[pre + "pooled_se"]).max() < 2e-5
        assert np.abs(o["bins_probs"].cpu().numpy() - g[pre + "bins_probs"]).max() < 2e-5
        assert np.array_equal(o["bins_mask"].bool().cpu().numpy(), g[pre + "bins_mask"])
        one = model.forward_raw(job[0], job[1], job[2], job[3], **cases.GRD_TH)
        for key in ("pooled_se", "bins_probs", "bins_mask"):
            assert torch.equal(one[key], o[key]), (v, key)


def test_grounding_batch_at_config_g_dims_equals_one_video_calls_and_splits_into_groups():
    """config G (d=128, 10 bins), 7 videos with T in [2, 125] and n in [1, 40]: batched == per video, bit for bit, also when the
    engine has to split the list into several calls"""
    cfg = dict(cases.GRD_CASES["grd_g_n64"]["cfg"])
    _, sd, _ = cases.make_grd_case("grd_g_n64")
    model = build_model(dict(cfg, loss_factor=dict(classification=1.0, centerness=1.0, regression=1.0)), sd)
    rs = np.random.RandomState(5)
    jobs = []
    for v, (T, n) in enumerate([(125, 40), (2, 3), (33, 1), (64, 17), (15, 40), (100, 9), (7, 5)]):
        feat, (quint, dura, vl) = cases.make_grd_inputs(cfg, dict(seed=900 + v, T=T, n=n, video_len=8 * T - int(rs.randint(0, 8))))
        jobs.append((feat.to(DEV), quint, dura, vl))
    ones = [model.forward_raw(*j, **cases.GRD_TH) for j in jobs]
    eng = model.engine()
    for cap in (eng.BATCH_ROWS, 3000):
        eng.BATCH_ROWS = cap
        outs = model.forward_raw_batch(jobs, **cases.GRD_TH)
        torch.cuda.synchronize()
        for v, (a, b) in enumerate(zip(ones, outs)):
            assert int(a["status"].item()) == int(b["status"].item())
            for key in ("pooled_se", "bins_probs", "bins_mask"):
                assert torch.equal(a[key], b[key]), (cap, v, key)


def test_head_chain_out_of_range_operands_fall_back_to_f32():
    """the LDS-resident three-head kernel (csrc/grounding_heads.hip) splits activations and weights into f16 pairs; an
    operand outside that form (|x| >= 32768, NaN, Inf) must not yield silent garbage.  (a) activations made huge by one scaled-up
    layer, (b) a point-wise weight beyond the range, (c) a NaN weight: the outputs must equal those of the exact-mode forward (the
    layer-by-layer f32 path, models/grd_model_v5.py:182-193 arithmetic), NaN included."""
    cfg = dict(cases.GRD_CASES["grd_g_n64"]["cfg"])
    _, sd0, _ = cases.make_grd_case("grd_g_n64")
    feat, (quint, dura, vl) = cases.make_grd_inputs(cfg, dict(seed=31, T=70, n=9, video_len=555))
    for case in ("huge_activation", "huge_weight", "nan_weight", "in_range"):
        sd = {k: v.clone() for k, v in sd0.items()}
        if case == "huge_activation":
            sd["conf_head.0.0.point_wise.weight"] *= 3e5          # layer 0's output ~1e5: the operand of layer 1 leaves the split form
            sd["conf_head.4.point_wise.weight"] *= 1e-5
        elif case == "huge_weight":
            sd["cls_head.2.0.point_wise.weight"][5, 7] = 5e4
            sd["cls_head.4.point_wise.weight"] *= 1e-4
        elif case == "nan_weight":
            sd["conf_head.1.0.point_wise.weight"][3, 3] = float("nan")
        model = build_model(dict(cfg, loss_factor=dict(classification=1.0, centerness=1.0, regression=1.0)), sd)
        got = model.forward_raw(feat.to(DEV), quint, dura, vl, keep_intermediates=True, **cases.GRD_TH)
        eng = model.engine()
        eng.set_mode(1, -1)
        try:
            want = model.forward_raw(feat.to(DEV), quint, dura, vl, keep_intermediates=True, **cases.GRD_TH)
        finally:
            eng.set_mode(-1, -1)
        if case == "nan_weight":
            # a NaN weight poisons every output of its head, as in torch (relu(NaN) = NaN); the other heads are untouched.  (The layer-by-
            # layer kernels clamp with max(x, 0), which drops a NaN at the next ReLU: no reference for this case.)
            assert torch.isnan(got["conf"]).all() and not torch.isnan(got["cls"]).any() and not torch.isnan(got["regrs"]).any()
            assert float((got["cls"] - want["cls"]).abs().max()) <= 2e-4 * float(want["cls"].abs().max())
            continue
        for key in ("regrs", "conf", "cls"):
            g, w = got[key].cpu(), want[key].cpu()
            assert torch.equal(torch.isnan(g), torch.isnan(w)), (case, key)
            ok = ~torch.isnan(w)
            scale = float(w[ok].abs().max()) if ok.any() else 1.0
            assert float((g[ok] - w[ok]).abs().max()) <= 2e-4 * max(scale, 1e-6), (case, key, float((g[ok] - w[ok]).abs().max()), scale)


def test_encoder_chain_out_of_range_operands_fall_back_to_f32():
    """The LDS-resident QANet encoder kernels (csrc/grounding_encoder.hip: convolution stack and the two GEMMs behind the attention) split
    their GEMM operands into f16 pairs as the head chain does: an operand outside that form (|x| >= 32768, NaN) must raise the flag that
    makes the f32 instantiation launched behind them redo the launch.  (a) a depth-wise weight that makes the point-wise operand huge,
    (b) a point-wise weight beyond the range, (c) attention values that make the operand of out_proj huge: the encoders' outputs equal
    those of the exact-mode forward (layer by layer in f32, models/grd_model_v5.py:110-137 arithmetic); (d) a NaN weight poisons its
    output feature in every row, as relu(NaN) + res does in torch."""
    cfg = dict(cases.GRD_CASES["grd_g_n64"]["cfg"])
    _, sd0, _ = cases.make_grd_case("grd_g_n64")
    feat, (quint, dura, vl) = cases.make_grd_inputs(cfg, dict(seed=33, T=70, n=9, video_len=557))
    d = cfg["dim_hidden"]
    for case in ("huge_depth_wise", "huge_weight", "huge_attention", "nan_weight", "in_range"):
        sd = {k: v.clone() for k, v in sd0.items()}
        if case == "huge_depth_wise":
            sd["combined_encoder.convs.1.depth_wise.weight"] *= 1e6
            sd["combined_encoder.convs.1.point_wise.weight"] *= 1e-6
        elif case == "huge_weight":
            sd["video_encoder.convs.2.point_wise.weight"][5, 7] = 5e4
        elif case == "huge_attention":
            sd["combined_encoder.mh_attn.in_proj_weight"][2 * d:] *= 1e6
            sd["combined_encoder.mh_attn.out_proj.weight"] *= 1e-6
        elif case == "nan_weight":
            sd["video_encoder.fc.weight"][3, 3] = float("nan")
        model = build_model(dict(cfg, loss_factor=dict(classification=1.0, centerness=1.0, regression=1.0)), sd)
        got = model.forward_raw(feat.to(DEV), quint, dura, vl, keep_intermediates=True, **cases.GRD_TH)
        if case == "nan_weight":
            assert torch.isnan(got["video_enc"][:, 3]).all() and not torch.isnan(got["query_enc"]).any()
            continue
        eng = model.engine()
        eng.set_mode(1, -1)
        try:
            want = model.forward_raw(feat.to(DEV), quint, dura, vl, keep_intermediates=True, **cases.GRD_TH)
        finally:
            eng.set_mode(-1, -1)
        for key in ("video_enc", "query_enc", "combined_enc", "regrs", "conf", "cls"):
            g, w = got[key].cpu(), want[key].cpu()
            assert not torch.isnan(g).any() and not torch.isnan(w).any(), (case, key)
            # (3e-4 of the largest magnitude: the out-of-range weight of `huge_weight` amplifies the f32 rounding of two different summation
            # orders -- the fallback's sequential FMAs against the exact mode's f32 MFMA -- 5e4-fold before the heads' sigmoids)
            assert float((g - w).abs().max()) <= 3e-4 * max(float(w.abs().max()), 1e-6), (case, key, float((g - w).abs().max()), float(w.abs().max()))


def test_encoder_chain_rows_do_not_depend_on_their_tile():
    """A row's result must not depend on where its sequence lies in the launch (tiles of the LDS-resident encoder kernels are cut from
    the row-major activation of ALL sequences): the same video scored alone, as first and as last of a batch -> identical bits."""
    cfg = dict(cases.GRD_CASES["grd_g_n64"]["cfg"])
    _, sd, _ = cases.make_grd_case("grd_g_n64")
    model = build_model(dict(cfg, loss_factor=dict(classification=1.0, centerness=1.0, regression=1.0)), sd)
    jobs = []
    for v, (T, n) in enumerate([(61, 7), (38, 11), (97, 3)]):
        feat, (quint, dura, vl) = cases.make_grd_inputs(cfg, dict(seed=700 + v, T=T, n=n, video_len=8 * T - 3))
        jobs.append((feat.to(DEV), quint, dura, vl))
    alone = model.forward_raw(*jobs[0], keep_intermediates=True, **cases.GRD_TH)
    for order in ([0, 1, 2], [2, 1, 0], [1, 0, 2]):
        outs = model.forward_raw_batch([jobs[i] for i in order], **cases.GRD_TH)
        o = outs[order.index(0)]
        for key in ("pooled_se", "bins_probs", "bins_mask"):
            assert torch.equal(o[key], alone[key]), (order, key)


_ENV_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ["BIGC_REPO"], "tests")); sys.path.insert(0, os.path.join(os.environ["BIGC_REPO"], "tests", "golden"))
import test_hip_grounding as T, cases
cfg = dict(cases.GRD_CASES["grd_g_n64"]["cfg"])
_, sd, _ = cases.make_grd_case("grd_g_n64")
model = T.build_model(dict(cfg, loss_factor=dict(classification=1.0, centerness=1.0, regression=1.0)), sd)
out = {}
for v, (Tn, n) in enumerate([(61, 7), (138, 11), (97, 30)]):
    feat, (quint, dura, vl) = cases.make_grd_inputs(cfg, dict(seed=800 + v, T=Tn, n=n, video_len=8 * Tn - 3))
    o = model.forward_raw(feat.to(T.DEV), quint, dura, vl, keep_intermediates=True, **cases.GRD_TH)
    for k in ("video_enc", "query_enc", "combined_enc", "regrs", "conf", "cls", "pooled_se", "bins_probs", "bins_mask"):
        out["%d_%s" % (v, k)] = o[k].cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_query_encoder_stream_and_fused_in_proj_against_the_plain_sequence():
    """Round 6: the query encoder runs on a second stream beside the video encoder (fork / join by events), the attention's in_proj runs
    inside the encoder chain kernel, the co-attention carries vq_fc, big k = 7 launches use 128-row tiles.  One stream (BIGC_GRD_STREAMS=0)
    and the forced tile size must give the same BITS; the in_proj / vq_fc as GEMMs of their own (other summation orders) the same values
    to the f32 level."""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tempfile.mkdtemp()
    res = {}
    for tag, env in (("default", {}), ("one_stream", {"BIGC_GRD_STREAMS": "0"}), ("plain_in_proj", {"BIGC_ENC_QKV": "0"}),
                     ("tile128", {"BIGC_ENC_TILE128": "1"}), ("two_kernel_coatt", {"BIGC_COATT_FC": "0"})):
        path = os.path.join(d, tag + ".npz")
        e = dict(os.environ, BIGC_REPO=repo, **env)
        r = subprocess.run([sys.executable, "-c", _ENV_SCRIPT, path], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = dict(np.load(path))
    for k, v in res["default"].items():
        assert np.array_equal(v, res["one_stream"][k], equal_nan=True), k
        # the k = 7 encoders on the 128-row tiles of big launches (forced here): the same arithmetic per row, the same bits
        assert np.array_equal(v, res["tile128"][k], equal_nan=True), k
        if v.dtype == np.bool_:
            continue
        assert rel(res["plain_in_proj"][k], v) < 5e-5, (k, rel(res["plain_in_proj"][k], v))
        # co-attention and vq_fc as two launches (the (rows, 4 d) tensor in memory, one GEMM over K = 4 d)
        assert rel(res["two_kernel_coatt"][k], v) < 5e-5, (k, rel(res["two_kernel_coatt"][k], v))


def test_grounding_forward_repeats_bit_for_bit_across_interleaved_batches():
    """The query encoder runs on the handle's second stream with its own scratch, joined into the caller's stream before the co-attention:
    two different batches scored alternately, twenty times, give their first results again every time (a missing dependency between the
    streams, or scratch shared by them, would show as differing bits)."""
    cfg = dict(cases.GRD_CASES["grd_g_n64"]["cfg"])
    _, sd, _ = cases.make_grd_case("grd_g_n64")
    model = build_model(dict(cfg, loss_factor=dict(classification=1.0, centerness=1.0, regression=1.0)), sd)
    batches = []
    for bi, shapes in enumerate([[(61, 7), (138, 11), (97, 30)], [(20, 40), (250, 3)]]):
        jobs = []
        for v, (T, n) in enumerate(shapes):
            feat, (quint, dura, vl) = cases.make_grd_inputs(cfg, dict(seed=1100 + 10 * bi + v, T=T, n=n, video_len=8 * T - 1))
            jobs.append((feat.to(DEV), quint, dura, vl))
        batches.append(jobs)
    first = [None, None]
    for it in range(20):
        bi = it & 1
        outs = model.forward_raw_batch(batches[bi], **cases.GRD_TH)
        got = [{k: o[k].clone() for k in ("pooled_se", "bins_probs", "bins_mask")} for o in outs]
        if first[bi] is None:
            first[bi] = got
            continue
        for v, (a, b) in enumerate(zip(first[bi], got)):
            for k in a:
                assert torch.equal(a[k], b[k]), (it, v, k)
