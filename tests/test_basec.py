"""The explicit pairwise baseline Base_C: CPU oracle and HIP path against outputs of the
reference's own ``models/model_pairwise_baseline.py`` (tests/golden/basec_*.npz)."""
import os
import tempfile

import numpy as np
import pytest
import torch

import cases
from oracle import basec_oracle as B


def torch_construct_triplet(proposal, pred_logits, topk, rt_triplets_topk):
    """(:306-395) with torch index operations on the device: the restatement the HIP kernels of csrc/pair_triplet.hip (used by
    ``forward``) are tested against; not on the product path."""
    dev = pred_logits.device
    n, k = proposal.num_proposals, topk
    probs = torch.softmax(pred_logits, dim=-1)
    scores, cats = torch.topk(probs, k, dim=-1)
    scores, cats = scores.reshape(-1), cats.reshape(-1)
    dura = proposal.traj_durations.to(dev)
    idx = torch.arange(n, device=dev)
    so = torch.stack(torch.meshgrid(idx, idx, indexing="ij"), -1).reshape(-1, 2)
    so = so[so[:, 0] != so[:, 1]]                                           # trajid2pairid (:105-112)
    so = torch.repeat_interleave(so, k, dim=0)
    s_d, o_d = dura[so[:, 0]], dura[so[:, 1]]
    start, end = torch.maximum(s_d[:, 0], o_d[:, 0]), torch.minimum(s_d[:, 1], o_d[:, 1])
    pos = end >= start                                                      # closed intervals (utils/utils_func.py:347-373)
    if int(pos.sum()) == 0:
        return None
    so, scores, cats, start, end = so[pos], scores[pos], cats[pos], start[pos], end[pos]
    enti_cat, enti_score = proposal.cat_ids.to(dev), proposal.scores.to(dev)
    quint = torch.cat([cats[:, None], enti_cat[so], so], dim=-1)            # [pred, s_cat, o_cat, s_tid, o_tid]
    trip = torch.stack([scores, enti_score[so[:, 0]], enti_score[so[:, 1]]], -1)
    # every (pair, predicate) occurs once: torch.unique(dim=0) only sorts the rows lexicographically (:349-358)
    key = (((quint[:, 0] * 512 + quint[:, 1]) * 512 + quint[:, 2]) * 1024 + quint[:, 3]) * 1024 + quint[:, 4]
    order = torch.argsort(key)
    quint, trip, inters = quint[order], trip[order], torch.stack([start, end], -1)[order]
    mask = quint[:, 0] != 0
    quint, trip, inters = quint[mask], trip[mask], inters[mask]
    if rt_triplets_topk > 0:
        top = trip.mean(dim=-1).argsort(descending=True)[:rt_triplets_topk]
        quint, trip, inters = quint[top], trip[top], inters[top]
    return quint, trip, inters, torch.empty(size=(trip.shape[0],))


def _check(res, g, i, tol):
    if bool(g["v%d_none" % i]):
        assert res is None
        return
    assert res is not None
    assert np.array_equal(res[0].cpu().numpy(), g["v%d_quint" % i])                 # bit-exact indices, incl. the top-k cut order
    assert np.array_equal(res[2].cpu().numpy(), g["v%d_dura" % i])
    want = g["v%d_scores" % i]
    got = res[1].cpu().numpy()
    assert np.array_equal(got[:, 1:], want[:, 1:])                                  # tracklet scores are copies
    assert np.abs(got[:, 0] - want[:, 0]).max() < tol


@pytest.mark.parametrize("name", list(cases.BASEC_CASES))
def test_basec_oracle_matches_reference(name):
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    cfg, sd, props, topk = cases.make_basec_case(name)
    for i, p in enumerate(props):
        if p.num_proposals > 1:
            logits, _ = B.forward_propagation(sd, cfg, p)
            ref = g["v%d_logits" % i]
            assert np.abs(logits.numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    for i, r in enumerate(B.forward_test(sd, cfg, props, topk)):
        _check(r, g, i, 1e-6)


@pytest.mark.parametrize("name", list(cases.BASEC_TRAIN_CASES))
def test_basec_train_oracle_matches_reference(name):
    """oracle restatement of _forward_train + loss (:276-312) with autograd vs the reference's loss, logits and gradients"""
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    cfg, sd, props, pos, labels = cases.make_basec_train_case(name)
    sdg = {k: (v.clone().requires_grad_(True) if k != "EntiNameEmb" else v) for k, v in sd.items()}
    total, terms, logits = B.forward_train(sdg, cfg, props, pos, labels)
    total.backward()
    assert abs(float(total) - float(g["total"])) < 1e-5 * abs(float(g["total"]))
    assert np.abs(logits.detach().numpy() - g["logits"]).max() <= 1e-5 * np.abs(g["logits"]).max()
    for k in g.files:
        if k.startswith("gnorm_"):
            got = sdg[k[6:]].grad
            assert abs(float(0.0 if got is None else got.norm()) - float(g[k])) <= 1e-4 * max(float(g[k]), 1e-8), k
        if k.startswith("grad_"):
            ref = g[k]
            assert np.abs(sdg[k[5:]].grad.numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-12), k


def _build(cfg, sd, is_train=False):
    from models import Base_C
    d = tempfile.mkdtemp()
    mc = dict(cfg)
    np.save(os.path.join(d, "b.npy"), sd["bias_matrix"].numpy())
    mc["bias_matrix_path"] = os.path.join(d, "b.npy")
    if "EntiNameEmb" in sd:
        np.save(os.path.join(d, "e.npy"), sd["EntiNameEmb"].numpy())
        mc["EntiNameEmb_path"] = os.path.join(d, "e.npy")
    m = Base_C(mc, is_train=is_train)
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.BASEC_CASES))
def test_basec_hip_matches_reference(name):
    from bigc_hip.packed import PackedBatch
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    cfg, sd, props, topk = cases.make_basec_case(name)
    model = _build(cfg, sd)
    dev_props = [p.to("cuda:0") for p in cases.make_basec_case(name)[2]]
    res = model(dev_props, topk=topk)
    assert len(res) == len(props)
    for i, r in enumerate(res):
        _check(r, g, i, 5e-5)
    # pair logits: 1e-4 relative (BASELINE north_star tolerance)
    keep = [p for p in dev_props if p.num_proposals > 0]
    out = model.engine().pair_forward(PackedBatch(keep, "cuda:0"))
    off, j = out["vid_pair_off"], 0
    for i, p in enumerate(props):
        if p.num_proposals == 0:
            continue
        if p.num_proposals > 1:
            ref = g["v%d_logits" % i]
            got = out["pair_logits"][off[j]:off[j + 1]].cpu().numpy()
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
        j += 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.BASEC_TRAIN_CASES))
def test_basec_train_hip_matches_reference(name):
    """HIP training step of Base_C: loss, logits of the labelled pairs and every parameter gradient vs the reference"""
    g = np.load(cases.GOLDEN_DIR + "/" + name + ".npz")
    cfg, sd, props, pos, labels = cases.make_basec_train_case(name)
    model = _build(cfg, sd, is_train=True)
    total, terms = model([p.to("cuda:0") for p in props], [x.to("cuda:0") for x in pos], [x.to("cuda:0") for x in labels])
    assert set(terms) == {"cls"}
    total.backward()
    torch.cuda.synchronize()
    assert abs(float(total) - float(g["total"])) < 2e-4 * abs(float(g["total"]))
    worst = {}
    for k, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None
            continue
        want = float(g["gnorm_" + k])
        got = p.grad
        assert got is not None, k
        worst[k] = abs(float(got.norm()) - want) / max(want, 1e-8)
        if "grad_" + k in g.files:
            ref = g["grad_" + k]
            worst[k] = max(worst[k], float(np.abs(got.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-12)))
    # same rule as test_hip_train.py: a ReLU at a pre-activation within rounding of zero may take the other branch
    bad = {k: v for k, v in worst.items() if v > 3e-3}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
    assert float(np.median(list(worst.values()))) < 1e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:10]


@pytest.mark.gpu
def test_basec_train_every_gradient_matches_oracle_and_optimizer_step_runs():
    name = "basec_train_tiny"
    cfg, sd, props, pos, labels = cases.make_basec_train_case(name)
    sdg = {k: (v.clone().requires_grad_(True) if k != "EntiNameEmb" else v) for k, v in sd.items()}
    total, _, logits = B.forward_train(sdg, cfg, props, pos, labels)
    (3.0 * total).backward()                                                   # a non-unit upstream gradient
    model = _build(cfg, sd, is_train=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    h_total, _ = model([p.to("cuda:0") for p in props], pos, labels)
    (3.0 * h_total).backward()
    for k, p in model.named_parameters():
        ref = sdg[k].grad
        if ref is None:
            continue
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= 2e-4 * max(float(ref.abs().max()), 1e-12), (k, err)
    before = float(h_total)
    for _ in range(5):                                                          # the reference loop body (tools/train_vidor.py:313-322)
        opt.zero_grad()
        loss, _ = model([p.to("cuda:0") for p in props], pos, labels)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=5, norm_type=2)
        opt.step()
    assert float(loss) < before
    # after the optimizer steps every derived buffer (split weights, transposed copies, the conv weight repacked for the backward)
    # must follow the CURRENT weights: gradients at the updated weights against the oracle's autograd at those weights
    sd2 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    sdg2 = {k: (v.requires_grad_(True) if k != "EntiNameEmb" else v) for k, v in sd2.items()}
    props_cpu = cases.make_basec_train_case(name)[2]                            # (.to("cuda:0") above moved `props` in place)
    total2, _, _ = B.forward_train(sdg2, cfg, props_cpu, pos, labels)
    total2.backward()
    opt.zero_grad()
    h2, _ = model([p.to("cuda:0") for p in props], pos, labels)
    h2.backward()
    assert abs(float(h2) - float(total2)) < 2e-4 * abs(float(total2))
    for k, p in model.named_parameters():
        ref = sdg2[k].grad
        if ref is None:
            continue
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= 1e-3 * max(float(ref.abs().max()), 1e-12), (k, err)


@pytest.mark.gpu
@pytest.mark.parametrize("n_trajs,rt,score_shift", [(180, 200, 0.0), (180, -1, 0.0), (37, 2048, 0.0), (2, 5, 0.0), (90, 150, -0.9), (60, 40, -3.0)])
def test_pair_triplets_kernel_vs_torch_restatement_at_vidor_size(n_trajs, rt, score_shift):
    """bigc_pair_triplets on random pair logits of a VidOR-sized video (N = 180: 32 220 pairs, 96 660 candidates, far beyond the
    LDS sort of triplet.hip) against the torch restatement of Base_C.construct_triplet (:314-395), which the reference goldens pin
    at N <= 24: rows, order, scores, spans -- with and without the rt_triplets_topk cut"""
    from bigc_hip.packed import PackedBatch
    from bigc_hip.synthetic import synthetic_proposal
    cfg, sd, _, _ = cases.make_basec_case("basec_tiny")
    cfg = dict(cfg, rt_triplets_topk=rt)
    model = _build(cfg, sd)
    P, topk = cfg["num_pred_cats"], 3
    model.topk = topk
    props = [synthetic_proposal(777, n_trajs=n_trajs, feat_width=36, num_enti_cats=cfg["num_enti_cats"], len_range=(5, 60), start_range=(0, 120)),
             synthetic_proposal(778, n_trajs=5, feat_width=36, num_enti_cats=cfg["num_enti_cats"], len_range=(5, 20), start_range=(0, 10))]
    for p in props:            # score_shift < 0: tracklet scores (and with them the mean scores of the top-R cut) of both signs, e.g. log-scores
        p.scores = p.scores + score_shift
    packed = PackedBatch([p.to("cuda:0") for p in props], "cuda:0")
    n = np.diff(packed.vid_trk_off_host)
    off = np.concatenate([[0], np.cumsum(n * (n - 1))])
    g = torch.Generator(device="cuda:0").manual_seed(5)
    logits = torch.randn((int(off[-1]), P), device="cuda:0", generator=g) * 2.0
    out = {"vid_pair_off": off, "pair_logits": logits, "_keepalive": torch.from_numpy(off.astype(np.int32)).to("cuda:0")}
    got = model.engine().pair_triplets(packed, out, topk, rt)
    for v, p in enumerate(props):
        want = torch_construct_triplet(p, logits[off[v]:off[v + 1]], model.topk, model.rt_triplets_topk)
        if want is None:
            assert got[v] is None
            continue
        q, s, d = got[v]
        assert torch.equal(q.cpu(), want[0].cpu()), "video %d quintuples" % v
        assert torch.equal(d.cpu(), want[2].cpu())
        assert float((s.cpu() - want[1].cpu()).abs().max()) < 1e-6


@pytest.mark.gpu
def test_pair_triplets_none_only_when_no_pair_overlaps():
    """models/model_pairwise_baseline.py:373-395: None when no ordered pair of the video overlaps in time; a video whose every row is
    the background predicate gives a tuple of EMPTY tensors"""
    from bigc_hip.packed import PackedBatch
    from bigc_hip.synthetic import synthetic_proposal
    cfg, sd, _, _ = cases.make_basec_case("basec_tiny")
    model = _build(dict(cfg, rt_triplets_topk=-1), sd)
    P = cfg["num_pred_cats"]
    model.topk = 1
    overlapping = synthetic_proposal(901, n_trajs=4, feat_width=36, num_enti_cats=cfg["num_enti_cats"], len_range=(20, 30), start_range=(0, 5))
    disjoint = synthetic_proposal(902, n_trajs=3, feat_width=36, num_enti_cats=cfg["num_enti_cats"], len_range=(5, 6), start_range=(0, 1))
    disjoint.traj_durations = torch.tensor([[0, 4], [10, 14], [20, 24]])
    props = [overlapping, disjoint]
    packed = PackedBatch([p.to("cuda:0") for p in props], "cuda:0")
    n = np.diff(packed.vid_trk_off_host)
    off = np.concatenate([[0], np.cumsum(n * (n - 1))])
    logits = torch.full((int(off[-1]), P), -5.0, device="cuda:0")
    logits[:, 0] = 5.0                                       # the background predicate wins every pair: with topk = 1 every row is dropped
    out = {"vid_pair_off": off, "pair_logits": logits, "_keepalive": torch.from_numpy(off.astype(np.int32)).to("cuda:0")}
    got = model.engine().pair_triplets(packed, out, 1, -1)
    assert got[1] is None
    assert got[0] is not None and got[0][0].shape == (0, 5) and got[0][1].shape == (0, 3) and got[0][2].shape == (0, 2)
