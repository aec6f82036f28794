"""GPU parity of the BIG-C eval forward (HIP path, through the C ABI) against
 (a) the committed golden outputs of the real reference and (b) the CPU oracle on seeded inputs.
Tolerance (BASELINE.json north_star): predicate logits within 1e-4 relative, top-K triplet
indices bit-exact."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cases
from bigc_hip import _lib
from bigc_hip.packed import PackedBatch
from bigc_hip.synthetic import synthetic_proposal
from oracle import bigc_oracle as O

DEV = "cuda:0"
LOGIT_RTOL = 1e-4
PROB_ATOL = 5e-5


def build_model(cfg, variant, sd):
    from models import BIG_C_vidvrd, BIG_C_vidor
    d = tempfile.mkdtemp()
    cfg = dict(cfg)
    np.save(os.path.join(d, "bias.npy"), sd["bias_matrix"].numpy())
    cfg["bias_matrix_path"] = os.path.join(d, "bias.npy")
    if "EntiNameEmb" in sd:
        np.save(os.path.join(d, "emb.npy"), sd["EntiNameEmb"].numpy())
        cfg["EntiNameEmb_path"] = os.path.join(d, "emb.npy")
    model = (BIG_C_vidvrd if variant == "vidvrd" else BIG_C_vidor)(cfg, is_train=False)
    model.load_state_dict(sd, strict=True)
    return model.to(DEV)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def check_logits(hip, golden, sd, cfg, variant, p):
    """north_star's bound on the predicate logits: within 1e-4 (max-norm, relative to the video's largest logit) of the REFERENCE's stored
    output -- or, where the two f32 evaluations are farther apart than that, within 1e-4 of a FLOAT64 evaluation of the oracle with the
    reference's own f32 output the farther of the two from it.  The seeded models' role attention is sharp by construction (att_gain 8:
    scores of magnitude ~50) and amplifies round-off; an f32 evaluation that rounds like the reference's is itself up to 2e-4 from float64
    on such videos (measured on the 256 benchmark videos; first seen at full dims on the exp1 case of round 6)."""
    e = rel(hip, golden)
    if e < LOGIT_RTOL:
        return
    from oracle import accuracy as A
    ref64 = A.f64_logits(sd, cfg, p, variant)
    e_hip, e_ref = A.rel_err(hip, ref64), A.rel_err(golden, ref64)
    assert e_hip <= LOGIT_RTOL and e_ref >= e_hip, "logits %.2e from the reference; from float64: HIP %.2e, the reference %.2e" % (e, e_hip, e_ref)


def check_tuple_exact(got, want_q, want_s, want_d, want_id):
    assert got is not None
    q, s, d, i = [t.cpu().numpy() for t in got]
    assert q.dtype == np.int64 and d.dtype == np.int64 and i.dtype == np.int64 and s.dtype == np.float32
    assert np.array_equal(q, want_q)
    assert np.array_equal(d, want_d)
    assert np.array_equal(i, want_id)
    assert np.array_equal(s[:, 1:], want_s[:, 1:])            # tracklet scores are copies
    assert np.abs(s[:, 0] - want_s[:, 0]).max() <= PROB_ATOL  # softmax of logits that agree to 1e-4 rel


@pytest.mark.parametrize("name", list(cases.BIGC_CASES))
def test_bigc_vs_reference_golden(name):
    """HIP forward vs the stored outputs of the real reference on the same seeded weights/inputs."""
    g = np.load(os.path.join(cases.GOLDEN_DIR, name + ".npz"))
    cfg, variant, sd, props, topk = cases.make_case(name)
    model = build_model(cfg, variant, sd)
    Q = cfg["num_querys"]
    # one video at a time (the reference's own batch size at test time) ...
    for v, p in enumerate(props):
        pre = "v%d_" % v
        res = model([p], topk=topk)
        assert len(res) == 1
        if bool(g[pre + "is_none"]):
            assert res[0] is None
        else:
            check_tuple_exact(res[0], g[pre + "quintuples"], g[pre + "scores"], g[pre + "dura_inters"], g[pre + "query_ids"])
        if p.num_proposals == 0:
            continue
        packed = PackedBatch([p], DEV)
        out = model.forward_packed(packed, topk, keep_intermediates=True)
        torch.cuda.synchronize()
        n = p.num_proposals
        if pre + "att_argmax" in g.files:            # outputs-only fixture: the contract itself
            check_logits(out["pred_logits"].cpu().numpy(), g[pre + "pred_logits"], sd, cfg, variant, p)
            assert np.array_equal(out["soid"].cpu().numpy()[0], g[pre + "att_argmax"])
            continue
        # intermediates, bounded at ~5 x what is measured (benchmarks/probes/intermediates_vs_golden.py, round 6: the encoder side agrees to
        # 9e-7 on every case; behind the sharp role attention -- scores of magnitude ~50 with att_gain 8, an f32 round-off of 1e-5 |s| moves a
        # probability by up to ~5e-4 -- att is within 2.4e-4 and pred_query within 1.05e-4 at full dims (N = 100), 7e-6 / 8e-6 at tiny dims).
        # The contract is on the logits (1e-4) and on the arg-max indices (exact).
        tiny = cfg["dim_enti"] <= 64
        assert rel(out["enti2enco"].cpu().numpy(), g[pre + "enti2enco"]) < 5e-6
        assert rel(out["enco_output"].cpu().numpy(), g[pre + "enco_output"]) < 5e-6
        assert rel(out["pred_query"].cpu().numpy(), g[pre + "pred_query"]) < (4e-5 if tiny else 2e-4)
        assert rel(out["att"].cpu().numpy().reshape(2, Q, n), g[pre + "att"]) < (4e-5 if tiny else 5e-4)
        check_logits(out["pred_logits"].cpu().numpy(), g[pre + "pred_logits"], sd, cfg, variant, p)
        assert np.array_equal(out["soid"].cpu().numpy()[0], g[pre + "att"].argmax(-1))
    # ... and all videos of the case in one batch (cross-video batching must not change anything)
    res = model(props, topk=topk)
    assert len(res) == len(props)
    for v, p in enumerate(props):
        pre = "v%d_" % v
        if bool(g[pre + "is_none"]):
            assert res[v] is None
        else:
            check_tuple_exact(res[v], g[pre + "quintuples"], g[pre + "scores"], g[pre + "dura_inters"], g[pre + "query_ids"])


def test_bigc_batch_vs_oracle_ragged():
    """A ragged batch (N from 1 to 40, L from 5 to 90, different Tmax per video) vs the CPU oracle."""
    cfg = cases.get_cfg("CONFIG_TINY")
    from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 21, att_gain=6.0)
    specs = [dict(seed=301, n_trajs=40, len_range=(5, 90), start_range=(0, 50)),
             dict(seed=302, n_trajs=1, len_range=(7, 8)),
             dict(seed=303, n_trajs=33, len_range=(5, 8), start_range=(0, 4)),
             dict(seed=304, n_trajs=0),
             dict(seed=305, n_trajs=17, len_range=(60, 64), start_range=(0, 2))]
    props = [cases.make_proposal(cfg, "vidvrd", s) for s in specs]
    model = build_model(cfg, "vidvrd", sd)
    got = model(props, topk=3)
    want = O.forward_test(sd, cfg, props, topk=3)
    assert len(got) == len(want)
    for gv, wv in zip(got, want):
        if wv is None:
            assert gv is None
            continue
        check_tuple_exact(gv, *[t.numpy() for t in wv])
    packed = PackedBatch([p for p in props if p.num_proposals > 0], DEV)
    out = model.forward_packed(packed, 3)
    torch.cuda.synchronize()
    logits = out["pred_logits"].cpu().numpy().reshape(packed.n_vid, cfg["num_querys"], -1)
    j = 0
    for p in props:
        if p.num_proposals == 0:
            continue
        o = O.encode2decode(sd, cfg, p)
        assert rel(logits[j], o["pred_logits"].numpy()) < LOGIT_RTOL
        j += 1


@pytest.mark.parametrize("variant,cfg_name,n_big", [("vidor", "CONFIG_TINY_O", 180), ("vidvrd", "CONFIG_TINY", 300)])
def test_bigc_batch_mixing_many_and_few_tracklets_vs_oracle(variant, cfg_name, n_big):
    """One batch holding a video with more than 64 tracklets (VidOR's 180; 300, beyond anything the datasets have) next to small ones:
    max_trk of the BATCH selects the unfused role-attention branch for every video of it, small ones included -- rows, durations and
    query ids identical to the oracle's per-video evaluation, logits within 1e-4."""
    cfg = cases.get_cfg(cfg_name)
    from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict
    sd = seeded_state_dict(bigc_state_shapes(cfg, variant), 23, att_gain=6.0)
    specs = [dict(seed=501, n_trajs=5, len_range=(5, 30), start_range=(0, 10)),
             dict(seed=502, n_trajs=n_big, len_range=(5, 60), start_range=(0, 40)),
             dict(seed=503, n_trajs=64, len_range=(6, 20), start_range=(0, 8)),
             dict(seed=504, n_trajs=2, len_range=(9, 12))]
    props = [cases.make_proposal(cfg, variant, s) for s in specs]
    model = build_model(cfg, variant, sd)
    topk = 3
    got = model(props, topk=topk)
    want = O.forward_test(sd, cfg, props, topk=topk, variant=variant)
    for gv, wv in zip(got, want):
        if wv is None:
            assert gv is None
            continue
        check_tuple_exact(gv, *[t.numpy() for t in wv])
    # ... and forward() batches a video only with videos of its own size class: every video's result is, bit for bit, what it gives alone
    for p_, gv in zip(props, got):
        alone = model([p_], topk=topk)[0]
        assert (alone is None) == (gv is None)
        if gv is not None:
            assert all(torch.equal(a, b) for a, b in zip(alone, gv))
    packed = PackedBatch(props, DEV)
    assert packed.max_trk == n_big
    out = model.forward_packed(packed, topk)
    logits = out["pred_logits"].cpu().numpy().reshape(packed.n_vid, cfg["num_querys"], -1)
    for j, p in enumerate(props):
        o = O.encode2decode(sd, cfg, p, variant)
        assert rel(logits[j], o["pred_logits"].numpy()) < LOGIT_RTOL
        assert np.array_equal(out["soid"].cpu().numpy()[j], o["att"].argmax(-1).numpy())


def test_construct_triplets_duplicates_and_ties():
    """a8 alone on crafted logits: duplicated quintuples across queries (same subject/object, same
    predicate), exact probability ties inside a query and between duplicates, background drop."""
    lib = _lib.load()
    rs = np.random.RandomState(5)
    nq, P, topk, N = 12, 9, 3, 6
    p = synthetic_proposal(401, n_trajs=N, feat_width=8, num_enti_cats=7, len_range=(20, 30), start_range=(0, 5))
    # distinct values inside a row (torch.topk's order among exactly tied probabilities is an
    # unspecified artefact of its partial sort -- probed: not lowest-index -- so ties inside a row are
    # not part of the contract), but the same few rows repeated across queries so that duplicate
    # quintuples with exactly tied probabilities exercise the first-maximum rule of :761
    base = np.stack([rs.permutation(P) for _ in range(3)]).astype(np.float32) * 0.5 - 2.0
    logits = torch.from_numpy(base[rs.randint(0, 3, size=nq)])
    logits[3] = logits[2]                                                              # identical queries
    att = torch.zeros(2, nq, N)
    so = rs.randint(0, N, size=(2, nq))
    so[:, 3] = so[:, 2]
    so[:, 5] = so[:, 4]
    for r in range(2):
        att[r, np.arange(nq), so[r]] = 1.0
    want = O.construct_triplet(p, logits, att, topk)
    e_off = torch.tensor([0, N], dtype=torch.int32, device=DEV)
    soid = torch.from_numpy(so.astype(np.int32)).reshape(1, 2, nq).to(DEV)
    dura = p.traj_durations.to(torch.int32).to(DEV)
    cat = p.cat_ids.to(torch.int32).to(DEV)
    score = p.scores.to(DEV)
    cap = nq * topk
    quint = torch.empty(1, cap, 5, dtype=torch.int32, device=DEV)
    scores = torch.empty(1, cap, 3, device=DEV)
    dd = torch.empty(1, cap, 2, dtype=torch.int32, device=DEV)
    qid = torch.empty(1, cap, dtype=torch.int32, device=DEV)
    n_out = torch.empty(1, dtype=torch.int32, device=DEV)
    ld = logits.to(DEV)
    _lib.check(lib.bigc_construct_triplets(_lib.current_stream(), ld.data_ptr(), P, soid.data_ptr(), e_off.data_ptr(), dura.data_ptr(),
                                           cat.data_ptr(), score.data_ptr(), 1, nq, topk, quint.data_ptr(), scores.data_ptr(),
                                           dd.data_ptr(), qid.data_ptr(), n_out.data_ptr()))
    torch.cuda.synchronize()
    n = int(n_out.item())
    assert want is not None and n == want[0].shape[0]
    assert np.array_equal(quint[0, :n].cpu().numpy(), want[0].numpy())
    assert np.array_equal(dd[0, :n].cpu().numpy(), want[2].numpy())
    assert np.array_equal(qid[0, :n].cpu().numpy(), want[3].numpy())
    assert np.abs(scores[0, :n].cpu().numpy() - want[1].numpy()).max() < 1e-6


def _run_triplets(p, logits, att, topk):
    lib = _lib.load()
    nq, P = logits.shape
    N = p.num_proposals
    so = att.argmax(-1).numpy()
    e_off = torch.tensor([0, N], dtype=torch.int32, device=DEV)
    soid = torch.from_numpy(so.astype(np.int32)).reshape(1, 2, nq).to(DEV)
    dura = p.traj_durations.to(torch.int32).to(DEV)
    cat = p.cat_ids.to(torch.int32).to(DEV)
    score = p.scores.to(DEV)
    cap = nq * topk
    quint = torch.empty(1, cap, 5, dtype=torch.int32, device=DEV)
    scores = torch.empty(1, cap, 3, device=DEV)
    dd = torch.empty(1, cap, 2, dtype=torch.int32, device=DEV)
    qid = torch.empty(1, cap, dtype=torch.int32, device=DEV)
    n_out = torch.empty(1, dtype=torch.int32, device=DEV)
    ld = logits.to(DEV)
    _lib.check(lib.bigc_construct_triplets(_lib.current_stream(), ld.data_ptr(), P, soid.data_ptr(), e_off.data_ptr(), dura.data_ptr(),
                                           cat.data_ptr(), score.data_ptr(), 1, nq, topk, quint.data_ptr(), scores.data_ptr(),
                                           dd.data_ptr(), qid.data_ptr(), n_out.data_ptr()))
    torch.cuda.synchronize()
    n = int(n_out.item())
    return quint[0, :n].cpu().numpy(), scores[0, :n].cpu().numpy(), dd[0, :n].cpu().numpy(), qid[0, :n].cpu().numpy()


def test_construct_triplets_exact_probability_ties_fixture():
    """G5: exact probability ties.  Ties INSIDE a query's top-k do not matter (the same k classes are kept, the
    output is sorted by quintuple).  A tie that straddles position k does: torch.topk's pick among exactly equal values is an
    artefact of its partial sort (tests/golden/bigc_ties.npz, produced by the reference: it is NOT the lowest index), the HIP path
    takes the lowest class ids.  Pinned here: (a) HIP == the oracle evaluated with the lowest-index rule, row for row; (b) on
    the queries without a straddling tie the reference keeps the same classes; (c) on the others the reference's picks are
    other members of the tied set."""
    g = np.load(os.path.join(cases.GOLDEN_DIR, "bigc_ties.npz"))
    p, logits, att, topk, boundary = cases.make_ties_case()
    got = _run_triplets(p, logits, att, topk)
    want = O.construct_triplet(p, logits, att, topk, topk_fn=O.topk_lowest_index)
    assert np.array_equal(got[0], want[0].numpy()) and np.array_equal(got[2], want[2].numpy()) and np.array_equal(got[3], want[3].numpy())
    assert np.abs(got[1] - want[1].numpy()).max() < 1e-6
    ours = O.topk_lowest_index(torch.softmax(logits, -1), topk)[1].numpy()
    ref = g["topk_classes"]
    differs = 0
    for q in range(logits.shape[0]):
        if q not in boundary:
            assert set(ours[q]) == set(ref[q]), q
            continue
        tied = set(boundary[q])
        forced = set(ours[q]) - tied
        assert forced == set(ref[q]) - tied and (set(ref[q]) - forced) <= tied, q
        assert sorted(set(ours[q]) - forced) == sorted(tied)[:topk - len(forced)], q          # lowest ids of the tied set
        differs += int(set(ours[q]) != set(ref[q]))
    assert differs >= 1            # the reference does not follow the lowest-index rule: the documented deviation is real


def test_wrong_feature_width_raises():
    cfg, variant, sd, props, topk = cases.make_case("bigc_tiny")
    model = build_model(cfg, variant, sd)
    bad = synthetic_proposal(1, n_trajs=3, feat_width=20, num_enti_cats=7, len_range=(5, 9))
    with pytest.raises(AssertionError):
        model([bad], topk=3)


def test_invalidate_weights_after_a_write_through_data():
    """parameters written through ``.data`` do not bump their version counter, which is what the engine watches: the documented
    remedy ``model.invalidate_weights()`` makes the next forward use the new values (and rebuild the split weight copies)"""
    cfg, variant, sd, props, topk = cases.make_case("bigc_tiny")
    model = build_model(cfg, variant, sd)
    packed = PackedBatch([props[0]], DEV)
    before = model.forward_packed(packed, topk)["pred_logits"].clone()
    with torch.no_grad():
        model.fc_pred2logits.weight.data.mul_(1.5)
        model.fc_enti2enco[0].weight.data.mul_(0.5)
    model.invalidate_weights()
    after = model.forward_packed(packed, topk)["pred_logits"].clone()
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["fc_pred2logits.weight"] *= 1.5
    sd2["fc_enti2enco.0.weight"] *= 0.5
    fresh = build_model(cfg, variant, sd2).forward_packed(packed, topk)["pred_logits"]
    assert not torch.equal(before, after)
    assert torch.equal(after, fresh)


def test_opt_in_fused_bbox_layers_reproduce_the_reference_goldens():
    """BIGC_PRE8_FUSED=1 (fc_bbox2enti.0 evaluated inside the staging of fc_bbox2enti.2, csrc/linear_split.hip): off by default
    because it is slower, but it has to stay correct -- the tiny and the full-dims goldens in a child process with the switch on"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BIGC_PRE8_FUSED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_hip_bigc.py", "-q", "-x", "-k",
                        "test_bigc_vs_reference_golden and (bigc_tiny or bigc_v_n64 or bigo_tiny_clsme)"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_chained_gemms_range_flag_reruns_the_step_on_the_exact_path():
    """exp2 dims, the default (chained) eval path: features scaled until a front-end activation no longer fits the split form
    (|x| >= 4096) -> the forward raises the range flag with the row counts, ``forward_checked`` repeats the step on the exact
    kernels, and the caller gets exactly what exact mode gives; the unscaled input does not raise the flag"""
    import copy
    from bigc_hip import _lib
    cfg, variant, sd, props, topk = cases.make_case("bigc_v_n64")
    model = build_model(cfg, variant, sd)
    p = props[0]
    packed = PackedBatch([copy.deepcopy(p).to(DEV)], DEV)
    out = model.forward_packed(packed, topk)
    assert int(out["range_flag"].item()) == 0
    big = copy.deepcopy(p)
    big.features_list = [f * 3000.0 for f in big.features_list]
    packed_big = PackedBatch([big.to(DEV)], DEV)
    raw = model.forward_packed(packed_big, topk)
    assert int(raw["range_flag"].item()) == 1
    with pytest.raises(OverflowError):
        model.counts(raw)
    got = model.forward_checked(packed_big, topk)
    lib = _lib.load()
    prev = lib.bigc_linear_set_exact(1)
    try:
        want = model.forward_packed(packed_big, topk)
        model.counts(want)
    finally:
        lib.bigc_linear_set_exact(prev)
    assert torch.equal(got["pred_logits"], want["pred_logits"]) and torch.equal(got["n_out"], want["n_out"])
    n = int(want["n_out"][0])      # rows past the count are padding, never written
    assert n > 0 and torch.equal(got["quint"][0, :n], want["quint"][0, :n]) and torch.equal(got["qid"][0, :n], want["qid"][0, :n])


@pytest.mark.parametrize("policy", [0, 1, 2, 3, 4])
def test_accumulation_policies_reproduce_the_reference_goldens(policy):
    """bigc_set_accum_policy (csrc/linear.hip; 4 = blocked accumulation in every pre-split GEMM, the default; 0 = one accumulator per element on
    the largest tiles): whatever the policy, the full-size reference goldens hold -- logits within 1e-4, attention arg-max and every triplet row
    identical -- and the policy in effect before is handed back.  A video's bits depend on the policy, not on its company: the same video
    alone and inside a batch of three gives identical logits under each policy."""
    lib = _lib.load()
    name = "bigc_v_n64"
    g = np.load(os.path.join(cases.GOLDEN_DIR, name + ".npz"))
    cfg, variant, sd, props, topk = cases.make_case(name)
    model = build_model(cfg, variant, sd)
    prev = lib.bigc_set_accum_policy(policy)
    try:
        assert lib.bigc_get_accum_policy() == policy
        p = props[0]
        res = model([p], topk=topk)
        check_tuple_exact(res[0], g["v0_quintuples"], g["v0_scores"], g["v0_dura_inters"], g["v0_query_ids"])
        out = model.forward_packed(PackedBatch([p], DEV), topk)
        alone = out["pred_logits"].clone()
        assert rel(alone.cpu().numpy(), g["v0_pred_logits"]) < LOGIT_RTOL
        assert np.array_equal(out["soid"].cpu().numpy()[0], np.argmax(g["v0_att"], -1))
        import copy
        trio = model.forward_packed(PackedBatch([copy.deepcopy(p).to(DEV) for _ in range(3)], DEV), topk)["pred_logits"]
        Q = cfg["num_querys"]
        assert torch.equal(trio[:Q], alone) and torch.equal(trio[2 * Q:3 * Q], alone)
    finally:
        assert lib.bigc_set_accum_policy(prev) == policy
