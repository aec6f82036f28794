"""Size-independent properties of the HIP path at the BASELINE workload's FULL dimensions (exp2: RoI 2048 + I3D 832, 2 encoder /
6 decoder layers, 192 queries, 133 predicates; videos of N=64 tracklets from the benchmark recipe), where the CPU oracle is
too slow to be the checker: determinism, independence of a video's result from its batch, permutation equivariance,
structural invariants of the triplets recomputed on the host from the inputs, subset relation between top-k settings,
agreement of the two GEMM modes on every triplet index, linearity of the GEMM."""
import numpy as np
import pytest
import torch

import cases
from bigc_hip import _lib, configs
from bigc_hip.packed import PackedBatch
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_proposal
from test_hip_bigc import build_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_VID = 12


@pytest.fixture(scope="module")
def setup():
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    model = build_model(cfg, "vidvrd", sd)
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    make = lambda: [synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"]) for v in range(N_VID)]
    return cfg, model, make


def as_np(res):
    return None if res is None else [t.cpu().numpy() for t in res]


def same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def test_deterministic_batch_independent_and_permutation_equivariant(setup):
    cfg, model, make = setup
    props = [p.to(DEV) for p in make()]
    r1 = [as_np(r) for r in model(props, topk=10)]
    r2 = [as_np(r) for r in model(props, topk=10)]
    assert all(same(a, b) for a, b in zip(r1, r2))                                   # run to run: bit-identical
    rev = [as_np(r) for r in model(props[::-1], topk=10)][::-1]
    assert all(same(a, b) for a, b in zip(r1, rev))                                  # order of the videos in the batch
    for i in (0, 5, N_VID - 1):                                                      # a video alone == the same video in the batch
        alone = as_np(model([props[i]], topk=10)[0])
        assert same(r1[i], alone)
    dup = [as_np(r) for r in model([props[3], props[3]], topk=10)]
    assert same(dup[0], dup[1]) and same(dup[0], r1[3])
    assert sum(r is not None for r in r1) == N_VID


def test_triplet_invariants_recomputed_on_the_host(setup):
    cfg, model, make = setup
    host = make()
    res = model([p.to(DEV) for p in make()], topk=10)
    for p, r in zip(host, res):
        q, s, d, qid = as_np(r)
        assert q.shape[0] == s.shape[0] == d.shape[0] == qid.shape[0] > 0
        assert (q[:, 0] > 0).all() and (q[:, 0] < cfg["num_pred_cats"]).all()        # background predicate removed
        assert (q[:, 3] != q[:, 4]).all()                                            # subject != object
        cats, sc, du = p.cat_ids.numpy(), p.scores.numpy(), p.traj_durations.numpy()
        assert np.array_equal(q[:, 1], cats[q[:, 3]]) and np.array_equal(q[:, 2], cats[q[:, 4]])
        assert np.array_equal(s[:, 1], sc[q[:, 3]]) and np.array_equal(s[:, 2], sc[q[:, 4]])
        assert ((s[:, 0] > 0) & (s[:, 0] <= 1)).all()
        lo = np.maximum(du[q[:, 3], 0], du[q[:, 4], 0])
        hi = np.minimum(du[q[:, 3], 1], du[q[:, 4], 1])
        assert np.array_equal(d[:, 0], lo) and np.array_equal(d[:, 1], hi) and (lo <= hi).all()   # closed-interval overlap
        key = [tuple(x) for x in q]
        assert len(set(key)) == len(key) and key == sorted(key)                      # unique, lexicographic (torch.unique(dim=0))
        assert ((qid >= 0) & (qid < cfg["num_querys"])).all()


def test_topk_subset_and_gemm_modes_agree(setup):
    cfg, model, make = setup
    props = [p.to(DEV) for p in make()]
    r10 = [as_np(r) for r in model(props, topk=10)]
    r3 = [as_np(r) for r in model(props, topk=3)]
    for a, b in zip(r3, r10):
        big = set(tuple(x) for x in b[0])
        assert all(tuple(x) in big for x in a[0])                                    # top-3 candidates are among the top-10
    # the two arithmetic modes (split-f16 MFMAs / exact f32 MFMAs) agree on every index, except where the decision itself is
    # a tie at f32 resolution: an arg-max over entities between two attention weights closer than 1e-4 relative (at most a
    # handful of the 2 * 192 * 12 decisions), or the arg-max over the queries that predicted the same triplet
    lib = _lib.load()
    packed = PackedBatch(props, torch.device(DEV))
    Q = cfg["num_querys"]

    def run():
        out = model.forward_packed(packed, 10)
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in out.items()}
    o_split = run()
    prev = lib.bigc_linear_set_exact(1)
    try:
        o_exact = run()
    finally:
        lib.bigc_linear_set_exact(prev)
    off = packed.vid_trk_off.cpu().numpy()
    flips = np.argwhere(o_split["soid"] != o_exact["soid"])
    assert len(flips) <= 6, len(flips)
    for v, r, q in flips:
        e0, ne = int(off[v]), int(off[v + 1] - off[v])
        row = o_exact["att"][2 * Q * e0 + (r * Q + q) * ne: 2 * Q * e0 + (r * Q + q + 1) * ne]
        a, b = row[o_split["soid"][v, r, q]], row[o_exact["soid"][v, r, q]]
        assert abs(a - b) <= 1e-4 * max(a, b), (v, r, q, a, b)
    clean = [v for v in range(N_VID) if v not in set(flips[:, 0].tolist())]
    assert len(clean) >= N_VID - 3
    # logits of the two modes: the f32-MFMA mode is itself 2e-4 away from a float64 evaluation (the split mode 3-5e-5,
    # tests/accuracy_vs_fp64.py), so the two may differ by a few 1e-4; queries whose arg-max entity differs are left out
    same_so = np.repeat((o_split["soid"] == o_exact["soid"]).all(1).reshape(-1), 1)
    diff = np.abs(o_split["pred_logits"] - o_exact["pred_logits"]).max(1)
    assert (diff * same_so).max() <= 6e-4 * np.abs(o_exact["pred_logits"]).max()
    for v in clean:
        n = int(o_exact["n_out"][v])
        assert n == int(o_split["n_out"][v])
        assert np.array_equal(o_split["quint"][v, :n], o_exact["quint"][v, :n]) and np.array_equal(o_split["dura"][v, :n], o_exact["dura"][v, :n])
        assert np.abs(o_split["scores"][v, :n] - o_exact["scores"][v, :n]).max() < 5e-5
        differ = o_split["qid"][v, :n] != o_exact["qid"][v, :n]
        assert differ.mean() <= 0.01, differ.sum()
        sa, sb = o_split["scores"][v, :n][differ, 0], o_exact["scores"][v, :n][differ, 0]
        assert np.all(np.abs(sa - sb) <= 2e-6 * np.abs(sb))


def test_gemm_linearity_at_full_size():
    """C(A1 + A2) = C(A1) + C(A2) - bias for the largest GEMM of the step (M = 579295, K = 2048, N = 512)"""
    lib = _lib.load()
    M, N, K = 579295, 512, 2048
    g = torch.Generator(device=DEV).manual_seed(5)
    A1 = torch.randn(M, K, device=DEV, generator=g)
    A2 = torch.randn(M, K, device=DEV, generator=g) * 0.5
    W = torch.randn(N, K, device=DEV, generator=g) * 0.02
    b = torch.randn(N, device=DEV, generator=g)

    def run(A):
        C = torch.empty(M, N, device=DEV)
        _lib.check(lib.bigc_linear_f32(_lib.current_stream(), A.data_ptr(), K, None, 1, K, W.data_ptr(), K, b.data_ptr(), None, N, C.data_ptr(), N,
                                       M, N, K, 0))
        return C
    c1, c2 = run(A1), run(A2)
    A1 += A2
    c12 = run(A1)
    err = (c12 - (c1 + c2 - b[None, :])).abs().max().item()
    assert err < 2e-5 * c12.abs().max().item()


def test_grounding_full_dims_invariants_and_batch_independence():
    """config G (1024-d clips, hidden 128, 10 bins) at n = 200 queries, T = 150 clips: determinism, a query's result does not
    depend on the other queries of the call, and the structural invariants of the pooled spans / masks."""
    from bigc_hip.synthetic import grd_state_shapes
    from test_hip_grounding import build_model as build_grd
    cfg = dict(configs.CONFIG_G)
    sd = seeded_state_dict(grd_state_shapes(cfg), 12, 4.0)
    model = build_grd(cfg, sd)
    rs = np.random.RandomState(3)
    n, T, vl = 200, 150, 1200
    feat = torch.from_numpy((0.1 * rs.standard_normal((T, cfg["dim_feat"]))).astype(np.float32)).to(DEV)
    quint = torch.from_numpy(np.stack([rs.randint(1, 51, n), rs.randint(1, 81, n), rs.randint(1, 81, n), rs.randint(0, 64, n), rs.randint(0, 64, n)], 1))
    start = rs.randint(0, vl - 100, n)
    dura = torch.from_numpy(np.stack([start, start + rs.randint(20, 100, n)], 1))
    th = dict(score_th=0.9, tiou_th=0.5, bins_th=0.2, nms_th=0.8)
    a = model([feat], [(quint, dura, vl)], with_gt_data=False, **th)
    b = model([feat], [(quint, dura, vl)], with_gt_data=False, **th)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    sub = model([feat], [(quint[40:90], dura[40:90], vl)], with_gt_data=False, **th)
    assert torch.equal(sub[2], a[2][40:90]) and torch.equal(sub[0], a[0][40:90]) and torch.equal(sub[1], a[1][40:90])
    pooled, probs, mask = (t.cpu().numpy() for t in a)
    k = cfg["num_bins"]
    assert pooled.shape == (n, k + 1, 2) and probs.shape == (n, k + 1) and mask.shape == (n, k + 1)
    assert (pooled >= 0).all() and (pooled <= 1).all() and (pooled[:, :, 0] <= pooled[:, :, 1]).all()
    assert (probs >= 0).all() and (probs <= 1).all()
    assert mask.any(axis=1).all()                                                    # at least one bin kept per query (:561-564)
    inter = dura.numpy() / vl
    assert np.abs(pooled[:, k, 0] - inter[:, 0]).max() < 1e-6 and np.abs(pooled[:, k, 1] - inter[:, 1]).max() < 1e-6   # bin k+1 = overlap span


def test_training_gradients_are_linear_in_the_upstream_gradient():
    """d(c * loss) = c * d(loss) through the whole HIP backward, at tiny dims (structure, not size, matters here)"""
    from test_hip_train import build_train_model
    cfg, variant, sd, props, gts = cases.make_train_case("bigc_train_tiny")
    grads = []
    for c in (1.0, 3.0):
        model = build_train_model(cfg, variant, sd)
        cfg2, _, _, props2, gts2 = cases.make_train_case("bigc_train_tiny")
        total, _ = model([p.to(DEV) for p in props2], gts2)
        (c * total).backward()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    for k in grads[0]:
        assert torch.allclose(grads[1][k], 3.0 * grads[0][k], rtol=2e-4, atol=1e-6 * float(grads[1][k].abs().max())), k   # fp32 atomics: summation order varies


def test_benchmark_videos_vs_oracle_row_for_row():
    """32 videos of the benchmark workload (bench.py: seeds 1234 + v, exp2 dims, N=64) through the HIP path in ONE batch and
    through the CPU oracle one by one: every triplet row (quintuple, overlap span, query id) identical, every attention arg-max
    identical, predicate logits within 1e-4 relative -- index parity at full size against the ORACLE (pinned to the reference by
    tests/golden/bigc_v_n64 / bigc_v_n100), not against the path's own second GEMM mode.

    PER VIDEO (round 5): the HIP logits are within 1e-4 of a FLOAT64 evaluation of the oracle, max-norm relative to max |logit| of the
    video -- asserted outright, no noise-floor criterion.  The benchmark model's attention is sharp by construction (scores of magnitude
    ~50) and amplifies round-off: an f32 evaluation whose GEMMs round like MKL's is up to 2.1e-4 from float64 on this set (the f32 oracle
    on video 29), and so was the HIP path while its GEMMs summed K in one accumulator (1.65e-4 on video 117, 1.59e-4 on 172).  With blocked
    accumulation in every GEMM (csrc/gemm_ps.hip) the maximum over all 256 benchmark videos is 6.7e-5 to 7.8e-5.  Where HIP and the f32
    oracle differ by more than 1e-4 the f32 oracle must be the farther one from float64.
    Over the SET, HIP's mean distance to float64 must not exceed 1.1 x the f32 oracle's (measured 0.7 - 0.8 x).
    PER QUERY ROW (round 6): the same bound with every query's own largest logit as the scale (6.0e-5 on these 32 videos, 8.7e-5 over
    all 256)."""
    from oracle import accuracy as A
    from oracle import bigc_oracle as O
    n_vid = 32
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    model = build_model(cfg, "vidvrd", sd)
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    Q, P = cfg["num_querys"], cfg["num_pred_cats"]
    host = [synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"]) for v in range(n_vid)]
    # packed as model(proposal_list) packs: nothing gathered yet, the forward reads the feature rows in place (the timed path of bench.py)
    packed = PackedBatch([synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"]).to(DEV) for v in range(n_vid)], DEV, "lazy")
    out = model.forward_packed(packed, 10)
    res = model.unpack_results(n_vid, list(range(n_vid)), out)
    logits = out["pred_logits"].cpu().numpy().reshape(n_vid, Q, P)
    soid = out["soid"].cpu().numpy()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    rows, arbitrated = 0, []
    e_hips, e_refs, row_errs = [], [], []
    with torch.no_grad():
        for v, p in enumerate(host):
            o = O.encode2decode(sd, cfg, p)
            want = O.construct_triplet(p, o["pred_logits"], o["att"], 10)
            got = as_np(res[v])
            assert (want is None) == (got is None)
            assert np.array_equal(soid[v], o["att"].argmax(-1).numpy()), "video %d: attention arg-max differs from the oracle" % v
            ref = o["pred_logits"].numpy()
            ref64 = A.f64_logits(sd, cfg, p, sd64=sd64)
            e_hips.append(A.rel_err(logits[v], ref64))
            e_refs.append(A.rel_err(ref, ref64))
            # north_star's bound, by arithmetic (round 5: blocked accumulation in every GEMM of the path): within 1e-4 of the float64
            # evaluation on EVERY video -- no noise-floor criterion.  Where two f32 evaluations are more than 1e-4 apart it is then the f32
            # oracle that sits far from float64 (video 29 of this set: 2.1e-4), which is recorded, not excused.
            assert e_hips[-1] <= 1e-4, "video %d: HIP logits %.2e from the float64 oracle (f32 oracle: %.2e)" % (v, e_hips[-1], e_refs[-1])
            # ... and PER QUERY ROW ("1e-4 rel" read row by row): max_c |delta| / max_c |logit of that query| against float64
            row_err = float((np.abs(logits[v].astype(np.float64) - ref64).max(1) / np.abs(ref64).max(1)).max())
            row_errs.append(row_err)
            assert row_err <= 1e-4, "video %d: a query row of the HIP logits is %.2e (relative to that row's largest logit) from float64" % (v, row_err)
            if np.abs(logits[v] - ref).max() / np.abs(ref).max() > 1e-4:
                arbitrated.append((v, dict(hip_vs_f64=e_hips[-1], f32_oracle_vs_f64=e_refs[-1])))
                assert e_refs[-1] > e_hips[-1], "video %d: HIP and the f32 oracle differ by more than 1e-4 and HIP is the farther from float64" % v
            if want is None:
                continue
            assert np.array_equal(got[0], want[0].numpy()), "video %d: quintuples differ" % v
            assert np.array_equal(got[2], want[2].numpy()) and np.array_equal(got[3], want[3].numpy())
            assert np.abs(got[1] - want[1].numpy()).max() <= 2e-4      # softmax probability of logits that agree to 1e-4 of max |logit|
            rows += want[0].shape[0]
    assert rows > 32 * 50 and len(arbitrated) <= 2, arbitrated
    assert max(e_hips) <= 1e-4 and np.mean(e_hips) <= 1.1 * np.mean(e_refs), "distance to float64: hip max %.2e mean %.2e, f32 oracle mean %.2e" % (
        max(e_hips), np.mean(e_hips), np.mean(e_refs))


def test_the_one_benchmark_row_that_differs_from_the_f32_oracle_is_the_float64_oracles():
    """Of the 51 306 triplet rows of the 256-video benchmark set ONE differs between the HIP path and the f32 oracle (video 156: a top-k
    decision between two class probabilities within round-off of each other).  Two f32 evaluations may disagree there; the float64
    evaluation of the oracle decides, and it sides with the HIP rows: every row of the HIP output that is not in the f32 oracle's output is
    in the float64 oracle's, at the same position.  (A video's bits do not depend on its company: scored alone here.)"""
    from oracle import accuracy as A
    from oracle import bigc_oracle as O
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    model = build_model(cfg, "vidvrd", sd)
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    v = 156
    p = synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
    got = as_np(model([synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"]).to(DEV)], topk=10)[0])
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        o = O.encode2decode(sd, cfg, p)
        want32 = O.construct_triplet(p, o["pred_logits"], o["att"], 10)
        _, p64 = A.to_f64({}, p)
        o64 = O.encode2decode(sd64, cfg, p64)
        want64 = O.construct_triplet(p64, o64["pred_logits"], o64["att"], 10)
    assert got[0].shape == tuple(want64[0].shape) == tuple(want32[0].shape)
    same32 = (got[0] == want32[0].numpy()).all(1) & (got[2] == want32[2].numpy()).all(1) & (got[3] == want32[3].numpy())
    same64 = (got[0] == want64[0].numpy()).all(1) & (got[2] == want64[2].numpy()).all(1) & (got[3] == want64[3].numpy())
    assert same64.all(), "rows %s of the HIP output differ from the float64 oracle's" % np.nonzero(~same64)[0].tolist()
    assert (~same32).sum() <= 1                      # (the f32 oracle's one tie that falls the other way; 0 if a library update moves it)
