"""Randomised differential test of the BIG-C eval forward: HIP path (through the C ABI) vs the CPU oracle on seeded random
CONFIGURATIONS -- widths of 32 .. 128 (so that the production kernels run: chained split-operand GEMMs, fused role attention
where a video has <= 64 tracklets, the unfused branch above), 1-2 encoder / decoder layers, 2-4 heads, 8-48 queries, with / without
I3D, the three VidOR head variants -- and random ragged batches (1 .. 90 tracklets of 5 .. 70 frames, empty videos, disjoint
tracklets).  Contract as everywhere: triplet rows, durations and query ids identical, logits within 1e-4 relative."""
import numpy as np
import pytest
import torch

import cases
from bigc_hip.packed import PackedBatch
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict
from oracle import bigc_oracle as O
from test_hip_bigc import LOGIT_RTOL, build_model, check_tuple_exact, rel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def random_case(seed):
    rs = np.random.RandomState(9000 + seed)
    variant = "vidvrd" if seed % 2 == 0 else "vidor"
    width = int(rs.choice([32, 64, 128]))
    heads = int(rs.choice([h for h in (2, 4, 8) if width // h in (8, 16, 32, 64)]))
    cfg = dict(num_enti_cats=int(rs.randint(3, 40)), num_pred_cats=int(rs.randint(3, 60)), dim_ffn=int(rs.choice([32, 64, 128])),
               dim_enti=width, dim_pred=width, dim_att=int(rs.choice([32, 64, 128])), dim_feat=int(rs.choice([32, 64, 96, 160])),
               dim_clsme=int(rs.choice([12, 20, 32])), enco_pool_len=4, n_enco_layers=int(rs.randint(1, 3)), n_deco_layers=int(rs.randint(1, 3)),
               n_att_head=heads, num_querys=int(rs.choice([8, 12, 20, 33, 48])), neg_weight=0.1, positive_vIoU_th=0.5,
               cost_coeff_dict=dict(classification=1.0, adj_matrix=30.0), loss_coeff_dict=dict(classification=1.0, adj_matrix=30.0),
               EntiNameEmb_path=None, bias_matrix_path=None)
    if variant == "vidvrd":
        if rs.rand() < 0.7:
            cfg["dim_i3d"] = int(rs.choice([16, 32, 48]))
    else:
        mode = int(rs.randint(0, 3))
        cfg["use_clsme"] = mode != 2
        if mode == 1:
            cfg["_with_EntiNameEmb"] = True
    n_vid = int(rs.randint(1, 6))
    specs = []
    for v in range(n_vid):
        kind = rs.rand()
        if kind < 0.1:
            specs.append(dict(seed=int(rs.randint(1 << 20)), n_trajs=0))
            continue
        n = int(rs.choice([1, 2, 5, 17, 40, 64, 65, 90]))
        lo = int(rs.randint(5, 30))
        specs.append(dict(seed=int(rs.randint(1 << 20)), n_trajs=n, len_range=(lo, lo + int(rs.randint(1, 40))), start_range=(0, int(rs.randint(1, 40))),
                          disjoint=bool(kind > 0.93)))
    return variant, cfg, specs, int(rs.randint(1, 6)), float(rs.choice([1.0, 4.0, 8.0]))


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_configuration_vs_oracle(seed):
    variant, cfg, specs, topk, gain = random_case(seed)
    topk = min(topk, cfg["num_pred_cats"] - 1)
    sd = seeded_state_dict(bigc_state_shapes(cfg, variant), 100 + seed, att_gain=gain)
    props = [cases.make_proposal(cfg, variant, s) for s in specs]
    model = build_model(cfg, variant, sd)
    got = model(props, topk=topk)
    want = O.forward_test(sd, cfg, props, topk=topk, variant=variant)
    assert len(got) == len(want) == len(props)
    for gv, wv in zip(got, want):
        if wv is None:
            assert gv is None
        else:
            check_tuple_exact(gv, *[t.numpy() for t in wv])
    # a video's result does not depend on its company in the batch: bit for bit what it gives alone
    for p_, gv in zip(props, got):
        alone = model([p_], topk=topk)[0]
        assert (alone is None) == (gv is None)
        if gv is not None:
            assert all(torch.equal(a, b) for a, b in zip(alone, gv)), (seed, p_.num_proposals)
    live = [p for p in props if p.num_proposals > 0]
    if not live:
        return
    out = model.forward_packed(PackedBatch(live, DEV), topk)
    logits = out["pred_logits"].cpu().numpy().reshape(len(live), cfg["num_querys"], -1)
    soid = out["soid"].cpu().numpy()
    for j, p in enumerate(live):
        o = O.encode2decode(sd, cfg, p, variant)
        assert rel(logits[j], o["pred_logits"].numpy()) < LOGIT_RTOL, (seed, j)
        assert np.array_equal(soid[j], o["att"].argmax(-1).numpy()), (seed, j)


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_grounding_configuration_vs_oracle(seed):
    """The grounding scorer on seeded random configurations (hidden width 64 / 128 -- 128 runs the streaming depth-wise kernels with
    the folded LayerNorm --, 2-10 bins, clip features of width 32-256) and random batches of videos (T = 1 .. 150 clips, n = 1 .. 30
    queries): the three heads within 1e-4 of the oracle's forward_propagation, and the batched call bit-identical to one-video calls."""
    from bigc_hip.synthetic import grd_state_shapes
    from oracle import grd_oracle as G
    from test_hip_grounding import build_model as build_grd
    rs = np.random.RandomState(7000 + seed)
    cfg = dict(dim_feat=int(rs.choice([32, 64, 256])), dim_clsme=300, dim_hidden=int(rs.choice([64, 128, 128])), num_bins=int(rs.randint(2, 11)),
               loss_factor=dict(classification=1.0, centerness=1.0, regression=1.0))
    sd = seeded_state_dict(grd_state_shapes(cfg), 300 + seed, float(rs.choice([1.0, 4.0])))
    model = build_grd(cfg, sd)
    k = cfg["num_bins"]
    jobs = []
    for v in range(int(rs.randint(1, 5))):
        spec = dict(seed=int(rs.randint(1 << 20)), T=int(rs.choice([1, 2, 3, 9, 40, 150])), n=int(rs.choice([1, 2, 7, 30])), video_len=int(rs.randint(20, 1200)))
        feat, (quint, dura, vl) = cases.make_grd_inputs(cfg, spec)
        jobs.append((feat, quint, dura, vl))
    th = dict(cases.GRD_TH)
    outs = model.forward_raw_batch([(f.to(DEV), q, d, vl) for f, q, d, vl in jobs], **th)
    for (feat, quint, dura, vl), o in zip(jobs, outs):
        n, T = quint.shape[0], feat.shape[0]
        one = model.forward_raw(feat.to(DEV), quint, dura, vl, keep_intermediates=True, **th)
        words, info = G.prepare_data(sd, quint, dura, vl)
        with torch.no_grad():
            regrs, conf, cls = G.forward_propagation(sd, feat, words, info)
        assert rel(one["regrs"].cpu().numpy().reshape(n, T, 2 * k), regrs.numpy()) < 1e-4
        assert rel(one["conf"].cpu().numpy().reshape(n, T, k), conf.numpy()) < 1e-4
        assert rel(one["cls"].cpu().numpy().reshape(n, T, k), cls.numpy()) < 1e-4
        for key in ("pooled_se", "bins_probs", "bins_mask"):
            assert torch.equal(one[key], o[key]), (seed, key)


@pytest.mark.parametrize("hidden", [64, 128])
def test_grounding_batch_mixing_short_and_long_videos_equals_one_video_calls(hidden):
    """The attention kernels switch from the single-pass to the two-pass form at a head-dim dependent sequence length (256 clips at
    head dims 8 / 16), per launch: the engine only batches videos on the same side of that limit, so a 40-clip video scored next to a
    300-clip one gets exactly the bits of its one-video call (found by the randomised test above at head dim 8, where the limit used
    to be 64 clips)."""
    from bigc_hip.synthetic import grd_state_shapes
    from test_hip_grounding import build_model as build_grd
    cfg = dict(dim_feat=64, dim_clsme=300, dim_hidden=hidden, num_bins=4, loss_factor=dict(classification=1.0, centerness=1.0, regression=1.0))
    sd = seeded_state_dict(grd_state_shapes(cfg), 77, 4.0)
    model = build_grd(cfg, sd)
    jobs = []
    for v, (T, n) in enumerate([(40, 5), (300, 3), (150, 4), (257, 2), (2, 6)]):
        feat, (quint, dura, vl) = cases.make_grd_inputs(cfg, dict(seed=880 + v, T=T, n=n, video_len=8 * T))
        jobs.append((feat.to(DEV), quint, dura, vl))
    th = dict(cases.GRD_TH)
    outs = model.forward_raw_batch(jobs, **th)
    assert len(outs) == len(jobs)
    for job, o in zip(jobs, outs):
        one = model.forward_raw(*job, **th)
        for key in ("pooled_se", "bins_probs", "bins_mask"):
            assert torch.equal(one[key], o[key]), (int(job[0].shape[0]), key)


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_training_step_vs_oracle_autograd(seed):
    """The BIG-C training step (forward, Hungarian matching on the device-computed cost, loss, hand-written backward) on seeded random
    configurations and batches against the oracle's autograd: total loss within 2e-4, matching identical, every parameter gradient
    within 3e-3 of the oracle's (relative to its largest entry) and the median parameter within 1e-4 -- the bounds of the golden tests
    (tests/test_hip_train.py explains the slack: a ReLU within f32 round-off of zero may take the other branch)."""
    import copy
    from bigc_hip.synthetic import synthetic_gt_graph
    from oracle import bigc_train_oracle as T
    from test_hip_train import build_train_model
    variant, cfg, specs, _, gain = random_case(50 + seed)
    specs = [s for s in specs if s.get("n_trajs", 0) >= 5 and not s.get("disjoint")][:3]
    if not specs:
        specs = [dict(seed=4242 + seed, n_trajs=12, len_range=(6, 30), start_range=(0, 10))]
    for s in specs:
        s["n_trajs"] = min(s["n_trajs"], 40)
        s["start_range"] = (0, 6)                                   # overlapping tracklets: the GT sampler needs overlapping pairs
    sd = seeded_state_dict(bigc_state_shapes(cfg, variant), 500 + seed, att_gain=min(gain, 3.0))
    props = [cases.make_proposal(cfg, variant, s) for s in specs]
    gts = [synthetic_gt_graph(p, 600 + seed + i, n_gt_trajs=min(5, p.num_proposals), n_preds=4, num_pred_cats=cfg["num_pred_cats"]) for i, p in enumerate(props)]
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want_total, _, want_idx = T.forward_train(sdg, cfg, copy.deepcopy(props), copy.deepcopy(gts), variant)
    want_total.backward()
    model = build_train_model(cfg, variant, sd)
    total, terms = model(props, copy.deepcopy(gts))
    total.backward()
    torch.cuda.synchronize()
    assert abs(float(total.detach()) - float(want_total.detach())) < 2e-4 * abs(float(want_total.detach()))
    for (iq, ig), (wq, wg) in zip(model.last_match_indices, want_idx):
        assert np.array_equal(np.asarray(iq), np.asarray(wq)) and np.array_equal(np.asarray(ig), np.asarray(wg))
    worst = {}
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        w_ = sdg[k].grad
        if w_ is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        worst[k] = float((p.grad.cpu() - w_).abs().max() / max(float(w_.abs().max()), 1e-12))
    bad = {k: v for k, v in worst.items() if v > 3e-3}
    assert not bad, (seed, sorted(bad.items(), key=lambda kv: -kv[1])[:8])
    assert float(np.median(list(worst.values()))) < 1e-4


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_pairwise_baseline_vs_oracle(seed):
    """Base_C (the explicit N x N pair classifier, models/model_pairwise_baseline.py) on seeded random configurations: triplet rows
    (incl. the order of the rt_triplets_topk cut), spans and tracklet scores identical to the oracle's, predicate probability within
    5e-5, pair logits within 1e-4."""
    from bigc_hip.synthetic import basec_state_shapes, synthetic_proposal
    from oracle import basec_oracle as B
    from test_basec import _build
    rs = np.random.RandomState(8000 + seed)
    width = int(rs.choice([16, 32, 64, 128]))
    mode = int(rs.randint(0, 3))
    cfg = dict(num_enti_cats=int(rs.randint(3, 40)), num_pred_cats=int(rs.randint(3, 50)), dim_feat=int(rs.choice([24, 64, 96])),
               dim_clsme=int(rs.choice([12, 20, 32])), dim_enti=width, dim_ffn=int(rs.choice([32, 64, 128])), enco_pool_len=4,
               use_clsme=mode != 2, rt_triplets_topk=int(rs.choice([-1, 5, 50, 200])), EntiNameEmb_path=None)
    if mode == 1:
        cfg["_with_EntiNameEmb"] = True
    sd = seeded_state_dict(basec_state_shapes(cfg), 900 + seed, 1.0)
    fw = cfg["dim_feat"] + (cfg["dim_clsme"] if (cfg["use_clsme"] and not cfg.get("_with_EntiNameEmb")) else 0)
    props = []
    for v in range(int(rs.randint(1, 5))):
        n = int(rs.choice([0, 1, 2, 7, 30, 70]))
        kw = dict(n_trajs=n, feat_width=fw, num_enti_cats=cfg["num_enti_cats"])
        if n:
            lo = int(rs.randint(4, 20))
            kw.update(len_range=(lo, lo + int(rs.randint(1, 30))), start_range=(0, int(rs.randint(1, 40))))
        props.append(synthetic_proposal(int(rs.randint(1 << 20)), **kw))
    topk = int(min(rs.randint(1, 5), cfg["num_pred_cats"] - 1))
    model = _build(cfg, sd)
    import copy
    got = model([copy.deepcopy(p).to(DEV) for p in props], topk=topk)
    want = B.forward_test(sd, cfg, props, topk=topk)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if w is None:
            assert g is None
            continue
        assert g is not None
        assert np.array_equal(g[0].cpu().numpy(), w[0].numpy()) and np.array_equal(g[2].cpu().numpy(), w[2].numpy())
        gs, ws = g[1].cpu().numpy(), w[1].numpy()
        assert np.array_equal(gs[:, 1:], ws[:, 1:]) and np.abs(gs[:, 0] - ws[:, 0]).max() < 5e-5
    for p_, g in zip(props, got):                          # ... and independent of the batch a video is scored in
        alone = model([copy.deepcopy(p_).to(DEV)], topk=topk)[0]
        assert (alone is None) == (g is None)
        if g is not None:
            # (the fourth entry is an uninitialised placeholder, as in the reference: models/model_pairwise_baseline.py:395)
            assert all(torch.equal(a, b) for a, b in zip(alone[:3], g[:3])), (seed, p_.num_proposals)
    keep = [copy.deepcopy(p).to(DEV) for p in props if p.num_proposals > 0]
    if keep:
        out = model.engine().pair_forward(PackedBatch(keep, DEV))
        off, j = out["vid_pair_off"], 0
        for p in props:
            if p.num_proposals == 0:
                continue
            if p.num_proposals > 1:
                ref = B.forward_propagation(sd, cfg, p)[0].numpy()
                got_l = out["pair_logits"][off[j]:off[j + 1]].cpu().numpy()
                assert got_l.shape == ref.shape and np.abs(got_l - ref).max() <= 1e-4 * np.abs(ref).max()
            j += 1


@pytest.mark.parametrize("seed", list(range(30, 38)))
def test_random_evaluator_device_tables_equal_host_viou(seed):
    """The evaluator with its vIoU overlaps from the batched device table (bigc_viou_pairs) against the per-pair host evaluation
    (= VidVRDhelperEvalAPIs/common.py:65-106, pinned by tests/golden/evaluator.npz) on seeded random GT / prediction sets:
    mAP, Recall@50 / @100 and the tagging precisions identical."""
    from bigc_hip import evaluate as E
    gt, pr = cases.make_eval_case(seed, n_videos=int(3 + seed % 5))
    m_h, r_h, p_h = E.evaluate(gt, pr)
    m_d, r_d, p_d = E.evaluate(gt, pr, device=DEV)
    assert abs(m_h - m_d) < 1e-9 and r_h == r_d and p_h == p_d


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_grounding_training_step_vs_oracle_autograd(seed):
    """The grounding scorer's training step on seeded random configurations (hidden 64 / 128, 2-10 bins, 1-3 videos of 4-60 clips,
    the reference's randperm negatives replayed through the global RNG): the five loss terms within 1e-4 and every parameter gradient
    within 5e-4 of the oracle's autograd -- the bounds of the golden tests (tests/test_hip_grd_train.py)."""
    from oracle import grd_train_oracle as GT
    from test_hip_grd_train import build_model as build_grd_train
    rs = np.random.RandomState(6000 + seed)
    name = "_fuzz_%d" % seed
    cases.GRD_TRAIN_CASES[name] = dict(
        cfg=dict(dim_feat=int(rs.choice([32, 64, 128])), dim_clsme=300, dim_hidden=int(rs.choice([64, 128])), num_bins=int(rs.randint(2, 11))),
        wseed=700 + seed, gain=float(rs.choice([1.0, 2.0])), rng_seed=int(rs.randint(1 << 20)), full=False,
        videos=[dict(seed=int(rs.randint(1 << 20)), T=int(rs.choice([4, 9, 25, 60])), n_gt_trajs=int(rs.randint(3, 7)), n_preds=int(rs.randint(2, 7)))
                for _ in range(int(rs.randint(1, 4)))])
    try:
        cfg, sd, feats, gts, rng_seed = cases.make_grd_train_case(name)
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        torch.manual_seed(rng_seed)
        total, terms, _ = GT.forward_train(sdg, cfg, feats, gts)
        total.backward()
        cfg, sd, feats, gts, rng_seed = cases.make_grd_train_case(name)
        model = build_grd_train(cfg, sd)
        torch.manual_seed(rng_seed)
        h_total, h_terms = model([f.to(DEV) for f in feats], gts)
        h_total.backward()
    finally:
        del cases.GRD_TRAIN_CASES[name]
    for k in terms:
        assert abs(float(h_terms[k].detach()) - float(terms[k].detach())) < 1e-4 * max(1.0, abs(float(terms[k].detach()))), (seed, k)
    for k, p in model.named_parameters():
        ref = sdg[k].grad
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= 5e-4 * max(float(ref.abs().max()), 1e-12), (seed, k, err, float(ref.abs().max()))


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_linear_shapes_vs_float64(seed):
    """bigc_linear_f32 on seeded random shapes: M from 1 to 3000, N from 4 to 700 (any multiple of 4: narrow and ragged last tiles),
    K from 4 to 1500 (ragged last K slab), padded pitches, bias / ReLU / sigmoid / residual, row gather over 1-3 segments with -1
    (zero) rows -- against float64 within 6 f32 round-offs of sum |a||w|; and the pre-split-operand GEMM (bigc_gemm_ps) on the shapes
    it supports (K a multiple of 32, N of 8) with its f32 and split outputs."""
    from bigc_hip import _lib
    from test_hip_gemm_ps import hr8_decode, repr_tol, to_hr8
    lib = _lib.load()
    rs = np.random.RandomState(5000 + seed)
    g = torch.Generator(device=DEV).manual_seed(seed)
    M = int(rs.choice([1, 2, 31, 33, 127, 129, 500, 3000]))
    N = 4 * int(rs.randint(1, 176))
    nseg = int(rs.choice([0, 0, 1, 2, 3]))
    seg = 32 * int(rs.randint(1, 12)) if nseg > 1 else 4 * int(rs.randint(1, 376))
    K = seg * max(nseg, 1)
    lda = seg + 4 * int(rs.randint(0, 3))
    n_src = M if nseg == 0 else int(rs.randint(1, 400))
    src = torch.randn(n_src, lda, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * 0.06
    bias = torch.randn(N, device=DEV, generator=g) if rs.rand() < 0.7 else None
    res = torch.randn(M, N, device=DEV, generator=g) if rs.rand() < 0.4 else None
    act = int(rs.choice([0, 1, 2]))
    idx = None
    if nseg:
        idx = torch.from_numpy(rs.randint(-1, n_src, size=(M, nseg)).astype(np.int32)).to(DEV)
        zero = torch.zeros(1, seg, device=DEV)
        A64 = torch.cat([torch.cat([src[:, :seg], zero])[idx[:, s].long()] for s in range(nseg)], 1).double()
    else:
        A64 = src[:, :seg].double()
    ldc = N + 4 * int(rs.randint(0, 3))
    C = torch.full((M, ldc), float("nan"), device=DEV)
    _lib.check(lib.bigc_linear_f32(_lib.current_stream(), src.data_ptr(), lda, _lib.ptr(idx), max(nseg, 1), seg, W.data_ptr(), K, _lib.ptr(bias),
                                   _lib.ptr(res), N, C.data_ptr(), ldc, M, N, K, act))
    torch.cuda.synchronize()
    pre = A64 @ W.double().t() + (bias.double() if bias is not None else 0)
    want = torch.relu(pre) if act == 1 else (torch.sigmoid(pre) if act == 2 else pre)
    if res is not None:
        want = want + res.double()
    scale = A64.abs() @ W.double().abs().t() + (bias.double().abs() if bias is not None else 0) + (res.double().abs() if res is not None else 0) + 1e-30
    got = C[:, :N].double()
    assert torch.isnan(C[:, N:]).all()                                            # the padding of the output pitch is untouched
    if act == 2:
        assert float((got - want).abs().max()) < 2e-6
    else:
        assert float(((got - want).abs() / scale).max()) < 6 * 2.0 ** -24, (M, N, K, nseg, act)
    # the same product on the split-operand GEMM where its shape constraints hold (dense A only)
    if nseg == 0 and K % 32 == 0 and N % 8 == 0 and lda == seg:
        As, flag0 = to_hr8(src)
        C2 = torch.full((M, N), float("nan"), device=DEV)
        Cs = torch.zeros((M, 4 * N), dtype=torch.uint8, device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        _lib.check(lib.bigc_gemm_ps(_lib.current_stream(), As.data_ptr(), 4 * K, 1, None, 1, W.data_ptr(), K, _lib.ptr(bias), _lib.ptr(res), N if res is not None else 0,
                                    C2.data_ptr(), N, Cs.data_ptr(), 4 * N, M, N, K, act, flag.data_ptr()))
        torch.cuda.synchronize()
        assert int(flag.item()) == 0
        if act == 2:
            assert float((C2.double() - want).abs().max()) < 2e-6
        else:
            assert float(((C2.double() - want).abs() / scale).max()) < 8 * 2.0 ** -24, (M, N, K, act)
        c64 = C2.double().cpu().numpy()
        assert (np.abs(hr8_decode(Cs, M, N) - c64) <= repr_tol(c64)).all()


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_ragged_attention_vs_float64(seed):
    """bigc_mha_core_f32 on seeded random ragged self-attention problems: head dims 8 / 16 / 32 / 64, 1-12 heads, 1-9 sequences of
    1 .. 500 rows (the launcher's kernel choice follows the LONGEST one: split-f16 single pass, f32 single pass, two passes), sharp
    and flat score distributions -- against a float64 softmax(q k^T / sqrt(hd)) v, 1e-5 of the largest output; and each sequence
    alone gives the bits it gets inside the batch whenever both launches fall on the same side of the single-pass limit."""
    import math
    from bigc_hip import _lib
    lib = _lib.load()
    rs = np.random.RandomState(4000 + seed)
    g = torch.Generator(device=DEV).manual_seed(seed)
    hd = int(rs.choice([8, 16, 32, 64]))
    H = int(rs.choice([1, 2, 8, 12]))
    d = hd * H
    lens = [int(rs.choice([1, 2, 31, 32, 33, 64, 100, 192, 193, 256, 257, 500])) for _ in range(int(rs.randint(1, 10)))]
    gain = float(rs.choice([0.3, 1.0, 3.0]))
    n = sum(lens)
    qkv = torch.randn(n, 3 * d, device=DEV, generator=g)
    qkv[:, :2 * d] *= gain
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    offd = torch.from_numpy(off).to(DEV)

    def run(x, o_, offs, n_seq, max_len):
        _lib.check(lib.bigc_mha_core_f32(_lib.current_stream(), x.data_ptr(), 3 * d, x.data_ptr() + 4 * d, 3 * d, x.data_ptr() + 8 * d, 3 * d,
                                         o_.data_ptr(), d, offs.data_ptr(), offs.data_ptr(), n_seq, max_len, H, hd, None))
    o = torch.full((n, d), float("nan"), device=DEV)
    run(qkv, o, offd, len(lens), max(lens))
    torch.cuda.synchronize()
    assert not torch.isnan(o).any()
    limit = 32 * {8: 8, 16: 8, 32: 6, 64: 6}[hd]
    for s in range(len(lens)):
        a, b = int(off[s]), int(off[s + 1])
        q, k, v = (qkv[a:b, i * d:(i + 1) * d].double().reshape(-1, H, hd).transpose(0, 1) for i in range(3))
        want = (torch.softmax(q @ k.transpose(1, 2) / math.sqrt(hd), -1) @ v).transpose(0, 1).reshape(-1, d)
        assert float((o[a:b].double() - want).abs().max() / want.abs().max()) < 1e-5, (seed, s, lens[s])
        if (lens[s] > limit) == (max(lens) > limit):
            one = torch.full((b - a, d), float("nan"), device=DEV)
            x1 = qkv[a:b].contiguous()
            run(x1, one, torch.tensor([0, b - a], dtype=torch.int32, device=DEV), 1, b - a)
            torch.cuda.synchronize()
            assert torch.equal(one, o[a:b]), (seed, s, lens[s], "alone != in the batch")
