"""The ragged pack of a batch (``bigc_pack_tracklets`` / ``bigc_pack_video_meta`` / ``bigc_compact_triplets``, csrc/ingest.hip,
csrc/triplet.hip; reference: the stacking inside models/model_0v10.py:391-430 over the containers of
dataloaders/dataloader_vidvrd.py:14-77): the one-launch gather is bit-exact against ``torch.cat`` over the tracklet tensors, the
split form it writes is byte for byte what ``bigc_split_rows_f32`` makes of the f32 rows, and ``forward(proposal_list)`` through it
returns what the f32-packed path returns."""
import os
import tempfile

import numpy as np
import pytest
import torch

import cases
from bigc_hip import _lib
from bigc_hip.packed import PackedBatch
from bigc_hip.synthetic import TrajProposal, bigc_state_shapes, seeded_state_dict, synthetic_proposal

DEV = "cuda:0"


def _foreign(p, dev, misalign=False, pitch=0):
    """the same video as the reference's container holds it: one separately allocated tensor per tracklet (optionally at an
    address that is not 16-byte aligned, or as a row-strided view)"""
    q = TrajProposal.__new__(TrajProposal)
    q.video_name, q.video_len, q.video_wh, q.num_proposals = p.video_name, p.video_len, p.video_wh, p.num_proposals
    q.cat_ids, q.scores, q.traj_durations = p.cat_ids.to(dev), p.scores.to(dev), p.traj_durations.to(dev)
    q.bboxes_list, q.features_list = [], []
    for b, f in zip(p.bboxes_list, p.features_list):
        b, f = b.to(dev), f.to(dev)
        if misalign:
            buf = torch.empty((f.numel() + 1,), dtype=torch.float32, device=dev)
            buf[1:].copy_(f.reshape(-1))
            f = buf[1:].view(f.shape)
        elif pitch:
            buf = torch.zeros((f.shape[0], f.shape[1] + pitch), dtype=torch.float32, device=dev)
            buf[:, :f.shape[1]].copy_(f)
            f = buf[:, :f.shape[1]]
        else:
            f = f.clone()
        q.bboxes_list.append(b.clone())
        q.features_list.append(f)
    q.dim_feat = p.dim_feat
    return q


def _videos(width, n=(7, 3, 12), seeds=(11, 12, 13), len_range=(3, 40)):
    return [synthetic_proposal(s, n_trajs=k, feat_width=width, len_range=len_range) for s, k in zip(seeds, n)]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["blocks", "foreign", "misaligned", "pitched"])
def test_gather_is_bit_exact_against_torch_cat(kind):
    width, dim_feat = 96 + 40, 96
    host = _videos(width)
    want_f = torch.cat([f for p in host for f in p.features_list], 0)
    want_b = torch.cat([b for p in host for b in p.bboxes_list], 0)
    if kind == "blocks":
        props = [p.to(DEV) for p in _videos(width)]
    else:
        props = [_foreign(p, DEV, misalign=kind == "misaligned", pitch=8 if kind == "pitched" else 0) for p in host]
    pk = PackedBatch(props, DEV, "lazy")
    assert pk._vectorized == (kind != "misaligned")
    pk.materialize("f32")
    assert torch.equal(pk.feats.cpu(), want_f) and torch.equal(pk.boxes.cpu(), want_b)
    assert torch.equal(pk.trk_cat.cpu(), torch.cat([p.cat_ids for p in host]).int())
    assert torch.equal(pk.trk_score.cpu(), torch.cat([p.scores for p in host]))
    assert torch.equal(pk.trk_dura.cpu(), torch.cat([p.traj_durations for p in host]).int())
    # the split form straight from the sources == the split of the gathered f32 rows, byte for byte; the other columns as f32
    pk2 = PackedBatch(props, DEV, "lazy")
    pk2.materialize("hr8", dim_feat)
    assert pk2._feats is None and pk2.feats_hr8.shape == (pk.n_rows, 4 * dim_feat)
    pk.materialize("hr8", dim_feat)                      # from its f32 rows (bigc_split_rows_f32)
    assert torch.equal(pk2.feats_hr8, pk.feats_hr8)
    assert torch.equal(pk2.extra.cpu(), want_f[:, dim_feat:]) and torch.equal(pk2.boxes.cpu(), want_b)
    assert int(pk2.feats_hr8_flag.item()) == 0
    # asked for later, the f32 rows are gathered from the sources again
    assert torch.equal(pk2.feats.cpu(), want_f)


@pytest.mark.gpu
def test_gather_raises_the_range_flag_and_handles_long_tracklets():
    width = 64
    host = _videos(width, n=(2, 5), seeds=(5, 6), len_range=(300, 700))
    host[1].features_list[3][17, 9] = 5000.0            # outside the split form's range (|x| >= 4096)
    want_f = torch.cat([f for p in host for f in p.features_list], 0)
    props = [p.to(DEV) for p in host]
    pk = PackedBatch(props, DEV, "lazy")
    pk.materialize("hr8", 64)
    assert int(pk.feats_hr8_flag.item()) == 1 and pk.extra is None
    assert torch.equal(pk.feats.cpu(), want_f)


def _model(cfg_name="CONFIG_TINY", seed=5):
    from models import BIG_C_vidvrd
    cfg = cases.get_cfg(cfg_name)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), seed, att_gain=4.0)
    d = tempfile.mkdtemp()
    np.save(d + "/b.npy", sd["bias_matrix"].numpy()); np.save(d + "/e.npy", sd["EntiNameEmb"].numpy())
    model = BIG_C_vidvrd(dict(cfg, bias_matrix_path=d + "/b.npy", EntiNameEmb_path=d + "/e.npy"), is_train=False)
    model.load_state_dict(sd)
    return cfg, model.to(DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name", ["CONFIG_TINY", "CONFIG_V"])
def test_forward_of_a_proposal_list_reads_the_rows_in_place(cfg_name, monkeypatch):
    """forward(proposal_list) makes NO copy of the features (a table of row addresses, csrc/linear_split.hip reads through it); the
    per-video tuples are the padded arrays' rows in the reference's dtypes; gathered into the split form instead (BIGC_FEAT_IN_PLACE=0)
    the batch gives, bit for bit, what an f32-packed batch gives; and the two forms agree within the logit tolerance"""
    cfg, model = _model(cfg_name)
    width = cfg["dim_feat"] + (cfg.get("dim_i3d") or 0)
    n = (9, 0, 17, 5) if cfg_name != "CONFIG_V" else (64, 33)
    host = [synthetic_proposal(40 + i, n_trajs=k, feat_width=width, num_enti_cats=cfg["num_enti_cats"]) for i, k in enumerate(n)]
    props = [p.to(DEV) for p in host]
    res = model(props, topk=3)
    live = [p for p in props if p.num_proposals > 0]
    eng = model.engine()
    if not eng.lib.bigc_forward_reads_hr8(eng.handle):
        pytest.skip("this configuration's forward reads packed f32 rows")
    pk = PackedBatch(live, DEV, "lazy")
    out = model.forward_packed(pk, 3)
    assert pk._feats is None and pk.feats_hr8 is None and pk.feat_row_ptr is not None        # nothing was copied
    cnt = model.counts(out).tolist()
    j = 0
    for p, r in zip(props, res):
        if p.num_proposals == 0:
            assert r is None
            continue
        k = cnt[j]
        if k == 0:
            assert r is None
        else:
            assert r[0].dtype == torch.int64 and r[2].dtype == torch.int64 and r[3].dtype == torch.int64 and r[1].dtype == torch.float32
            assert torch.equal(r[0], out["quint"][j, :k].long()) and torch.equal(r[1], out["scores"][j, :k])
            assert torch.equal(r[2], out["dura"][j, :k].long()) and torch.equal(r[3], out["qid"][j, :k].long())
        j += 1
    # the split form: gathered straight from the sources == made from packed f32 rows
    pk_f32 = PackedBatch(live, DEV)
    out_f32 = model.forward_packed(pk_f32, 3)
    monkeypatch.setenv("BIGC_FEAT_IN_PLACE", "0")
    pk_hr8 = PackedBatch(live, DEV, "lazy")
    out_hr8 = model.forward_packed(pk_hr8, 3)
    assert pk_hr8._feats is None and pk_hr8.feats_hr8 is not None and pk_hr8.feat_row_ptr is None
    assert torch.equal(out_hr8["pred_logits"], out_f32["pred_logits"]) and model.counts(out_hr8).tolist() == model.counts(out_f32).tolist()
    for j, k in enumerate(model.counts(out_f32).tolist()):                # (rows past a video's count are padding, never written)
        assert torch.equal(out_hr8["quint"][j, :k], out_f32["quint"][j, :k]) and torch.equal(out_hr8["scores"][j, :k], out_f32["scores"][j, :k])
    # in place (fc_feat2enti.0 splits the rows in its loop, two accumulators) vs the split form (one accumulator): the same numbers to 1e-4
    Q = cfg["num_querys"]
    a, b = out["pred_logits"].reshape(len(live), Q, -1), out_f32["pred_logits"].reshape(len(live), Q, -1)
    same = (out["soid"] == out_f32["soid"]).all(1)                       # (B, Q): queries whose attention arg-max agrees
    rel = ((a - b).abs().amax(2) * same).amax(1) / b.abs().amax((1, 2))
    assert float(rel.max()) < 1e-4 and float(same.float().mean()) > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("in_place", ["1", "0"])
def test_feature_outside_the_split_forms_range_through_the_api(in_place, monkeypatch):
    """a feature of 6000: outside the range of the STORED split form (|x| >= 4096: the gather raises the flag, the step is repeated on the
    exact kernels and returns exactly what exact mode gives), inside the range of the in-loop split of the in-place path (|x| < 32768: the
    result is a valid split-arithmetic result, close to the exact mode's)"""
    monkeypatch.setenv("BIGC_FEAT_IN_PLACE", in_place)
    cfg, model = _model("CONFIG_V")
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    host = [synthetic_proposal(77, n_trajs=12, feat_width=width, num_enti_cats=cfg["num_enti_cats"])]
    host[0].features_list[2][3, 100] = 6000.0
    props = [p.to(DEV) for p in host]
    res = model(props, topk=3)
    eng = model.engine()
    eng.set_mode(1, -1)
    try:
        want = model(props, topk=3)
    finally:
        eng.set_mode(-1, -1)
    if in_place == "0":
        for a, b in zip(res[0], want[0]):
            assert torch.equal(a, b)
    else:
        assert torch.equal(res[0][0], want[0][0]) and torch.equal(res[0][2], want[0][2])
        assert float((res[0][1] - want[0][1]).abs().max()) < 1e-4


@pytest.mark.gpu
def test_forward_async_returns_what_forward_returns_with_several_batches_in_flight():
    """model.forward_async queues a batch and returns; .result() later gives, bit for bit, what model(proposal_list) gives -- also with three
    batches queued before the first result is fetched, batches of mixed size classes, an empty video, and a batch whose range flag is raised"""
    cfg, model = _model("CONFIG_V")
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    batches = []
    for b, sizes in enumerate([(20, 0, 70), (64,), (5, 9, 31, 12)]):
        host = [synthetic_proposal(300 + 10 * b + i, n_trajs=k, feat_width=width, num_enti_cats=cfg["num_enti_cats"]) for i, k in enumerate(sizes)]
        batches.append([p.to(DEV) for p in host])
    batches[2][1].features_list[0][0, 7] = 5.0e4          # |x| >= 32768: the in-loop split's own f32 fallback handles it (no flag), results stay valid
    pend = [model.forward_async(b, topk=3) for b in batches]
    got = [p.result() for p in pend]
    for b, g in zip(batches, got):
        want = model(b, topk=3)
        assert len(g) == len(want)
        for x, y in zip(g, want):
            assert (x is None) == (y is None)
            if x is not None:
                assert all(torch.equal(u, v) for u, v in zip(x, y))


@pytest.mark.gpu
def test_forward_async_refuses_a_ninth_pending_forward():
    """the row offsets of a pending forward wait in one of eight pinned buffers: a ninth pending forward would overwrite the first one's"""
    cfg, model = _model("CONFIG_V")
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    batch = [synthetic_proposal(500, n_trajs=6, feat_width=width, num_enti_cats=cfg["num_enti_cats"]).to(DEV)]
    want = model(batch, topk=3)
    pend = [model.forward_async(batch, topk=3) for _ in range(8)]
    with pytest.raises(AssertionError):
        model.forward_async(batch, topk=3)
    for p in pend:
        got = p.result()
        assert all(torch.equal(u, v) for u, v in zip(got[0], want[0]))
    model.forward_async(batch, topk=3).result()          # the slots are free again


class _RefStyleProposal(object):
    """A container that behaves like the reference's ``TrajProposal`` (dataloaders/dataloader_vidvrd.py:14-77): one tensor per tracklet
    in plain lists, and ``.to(device)`` REPLACES the list items one by one and returns self."""

    def __init__(self, p):
        self.video_name, self.video_len, self.video_wh, self.num_proposals = p.video_name, p.video_len, p.video_wh, p.num_proposals
        self.cat_ids, self.scores, self.traj_durations = p.cat_ids.clone(), p.scores.clone(), p.traj_durations.clone()
        self.bboxes_list = [b.clone() for b in p.bboxes_list]
        self.features_list = [f.clone() for f in p.features_list]
        self.dim_feat = p.dim_feat

    def to(self, device):
        for i in range(self.num_proposals):
            self.bboxes_list[i] = self.bboxes_list[i].to(device)
            self.features_list[i] = self.features_list[i].to(device)
        self.cat_ids, self.scores, self.traj_durations = self.cat_ids.to(device), self.scores.to(device), self.traj_durations.to(device)
        return self


@pytest.mark.gpu
def test_eval_loop_over_reference_style_containers_does_not_retain_gpu_features():
    """the eval tools move a video to the GPU, queue its forward, then ``p.to(cpu)`` -- and keep every proposal alive.  The
    source table cached on the proposal used to hold the per-tracklet GPU tensors, so every evaluated video's features stayed allocated
    until the run ended.  The cache holds addresses only now: allocated memory stays flat over the loop, and a video moved back to the
    GPU (new addresses) is read correctly."""
    cfg, model = _model("CONFIG_V")
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    vids = [_RefStyleProposal(synthetic_proposal(700 + i, n_trajs=24, feat_width=width, num_enti_cats=cfg["num_enti_cats"], len_range=(20, 40))) for i in range(6)]
    per_video = sum(f.numel() * 4 for f in vids[0].features_list)
    assert per_video > 2e6
    # warm-up (workspace, weights): one video through the whole cycle
    want0 = model([vids[0].to(DEV)], topk=3)[0]
    want0 = tuple(t.cpu() for t in want0)
    vids[0].to("cpu")
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    pend = None
    for p in vids:
        nxt = (p, model.forward_async([p.to(DEV)], topk=3))
        if pend is not None:
            pend[1].result()
            pend[0].to("cpu")
        pend = nxt
    pend[1].result()
    pend[0].to("cpu")
    del pend, nxt
    torch.cuda.synchronize()
    grown = torch.cuda.memory_allocated() - base
    assert grown < per_video, "GPU memory grew by %.1f MB over 6 videos of %.1f MB each" % (grown / 1e6, per_video / 1e6)
    assert all(not f.is_cuda for p in vids for f in p.features_list)
    # the first video once more on the GPU: fresh allocations, the stale table (other addresses) must not be used
    got = model([vids[0].to(DEV)], topk=3)[0]
    assert all(torch.equal(u.cpu(), v) for u, v in zip(got, want0))


def test_source_table_cache_is_validated_item_by_item():
    """a cache hit used to be decided by the lists' length and the addresses of their FIRST and LAST tensors.  After a
    to(GPU) -> forward -> to(cpu) -> to(GPU) cycle the caching allocator may hand back the old blocks for the ends and other blocks for
    the middle: the stale addresses were then read.  The cache holds a weak reference per source tensor now; a hit needs every one alive
    and identical to the list's current item.  (CPU tensors: the table logic is device-independent.)"""
    from bigc_hip.packed import tracklet_table
    dev = torch.device("cpu")
    p = _RefStyleProposal(synthetic_proposal(811, n_trajs=9, feat_width=16, len_range=(4, 9)))
    t1 = tracklet_table(p, dev)
    assert t1["fptr"].tolist() == [f.data_ptr() for f in p.features_list]
    t2 = tracklet_table(p, dev)
    assert t2["_cached"] is p._bigc_pack and t2["fptr"] is t1["fptr"]                     # unchanged lists: a hit
    # same ends (the very same tensor objects), middle tracklets replaced by tensors elsewhere in memory
    keep_old = [p.features_list[i] for i in range(2, 6)]                                   # (keeps the old blocks from being re-used for the clones)
    for i in range(2, 6):
        p.features_list[i] = p.features_list[i].clone()
    t3 = tracklet_table(p, dev)
    assert t3["fptr"].tolist() == [f.data_ptr() for f in p.features_list] and t3["fptr"].tolist() != t1["fptr"].tolist()
    # the same ADDRESS everywhere but other tensor objects (what allocator re-use looks like): not a hit either -- the table is rebuilt
    # from the lists as they are, which costs time, never correctness
    p.features_list = [f[:] for f in p.features_list]
    t4 = tracklet_table(p, dev)
    assert t4["fptr"] is not t3["fptr"] and t4["fptr"].tolist() == t3["fptr"].tolist()
    # a source tensor that died (its list item replaced and nothing else holding it): the weak reference is dead, no hit
    p.bboxes_list[4] = p.bboxes_list[4].clone()
    t5 = tracklet_table(p, dev)
    assert t5["bptr"].tolist() == [b.data_ptr() for b in p.bboxes_list]
    # what PackedBatch derives from the table is kept on the proposal's cached dict, not on the per-call copy
    from bigc_hip.packed import PackedBatch
    PackedBatch([p], "cpu")
    assert p._bigc_pack["tmax"] == max(b.shape[0] for b in p.bboxes_list)
    del keep_old


def _same_results(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert (x is None) == (y is None)
        if x is not None:
            assert all(torch.equal(u, v) for u, v in zip(x, y))


@pytest.mark.gpu
def test_same_list_again_reuses_the_tables_and_sees_values_written_in_place():
    """``forward(proposal_list)`` keeps its last batch; when the same containers come back with the same tensors
    (identity, element by element) the address / offset tables are not rebuilt -- but everything the batch COPIES (boxes, scores,
    categories, durations) is gathered again, so that values written in place between the calls are seen, as the feature rows (read in
    place) are.  Replaced tensors, a replaced container or another device drop the kept batch."""
    cfg, model = _model("CONFIG_V")
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    mk = lambda s: synthetic_proposal(s, n_trajs=12, feat_width=width, num_enti_cats=cfg["num_enti_cats"], len_range=(10, 30))
    props = [mk(900 + i).to(DEV) for i in range(5)]
    first = model(props, topk=3)
    assert model._last_pack["packed"] is None                               # a first call only remembers (length, first container) ...
    _same_results(model(props, topk=3), first)
    kept = model._last_pack["packed"]                                       # ... the second call with the same first container keeps its batch ...
    assert kept is not None and kept.proposals is None                      # (weak references only: the batch does not hold the containers)
    _same_results(model(props, topk=3), first)
    assert model._last_pack["packed"] is kept                               # ... and the third is served by it
    _same_results(model(list(props), topk=3), first)                        # another list object with the same containers: still a hit
    assert model._last_pack["packed"] is kept
    # values written IN PLACE: features (read in place), scores and boxes (gathered again)
    with torch.no_grad():
        props[1].features_list[0].mul_(1.5)
        props[2].scores.mul_(0.5)
        props[3].bboxes_list[2].add_(3.0)
    got = model(props, topk=3)
    assert model._last_pack["packed"] is kept
    fresh_cfg, fresh = _model("CONFIG_V")
    want = fresh([p for p in props], topk=3)                                # a model that has never seen the list
    _same_results(got, want)
    assert not all(torch.equal(a[1], b[1]) for a, b in zip(got, first) if a is not None and b is not None)
    # a container whose tensors were replaced (to(cpu) -> to(GPU)): rebuilt (and, the first container being the same, kept again)
    props[0].to("cpu").to(DEV)
    _same_results(model(props, topk=3), want)
    assert model._last_pack["packed"] is not kept and model._last_pack["packed"] is not None
    kept = model._last_pack["packed"]
    # a metadata tensor replaced by another object
    props[4].scores = props[4].scores.clone()
    _same_results(model(props, topk=3), want)
    assert model._last_pack["packed"] is not kept
    # a shorter list, then a foreign container in the list: no reuse, no stale batch kept
    _same_results(model(props[:3], topk=3), want[:3])
    assert model._last_pack["packed"] is None
    mixed = props[:4] + [_foreign(mk(904), DEV)]
    for _ in range(3):
        r = model(mixed, topk=3)
        _same_results(r[:4], want[:4])
        assert model._last_pack["packed"] is None                           # (a batch with a foreign container is never kept)
    # ever new lists: nothing but (length, first container) is remembered
    for i in range(3):
        model([mk(960 + i).to(DEV)], topk=3)
        assert model._last_pack["packed"] is None


@pytest.mark.gpu
def test_lazy_result_list_behaves_like_the_reference_list():
    """the eval result of a one-class batch is a ``list`` whose entries are cut out of the compacted arrays when first read"""
    cfg, model = _model("CONFIG_V")
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    props = [synthetic_proposal(950 + i, n_trajs=9, feat_width=width, num_enti_cats=cfg["num_enti_cats"], len_range=(10, 30)).to(DEV) for i in range(4)]
    res = model(props, topk=3)
    assert isinstance(res, list) and len(res) == 4
    eager = model.unpack_results(4, [0, 1, 2, 3], model.forward_checked(PackedBatch(props, DEV, "lazy"), 3))
    assert type(eager) is list
    _same_results(res, eager)                      # iteration
    _same_results([res[i] for i in range(4)], eager)
    _same_results(res[1:3], eager[1:3])
    _same_results([res[-1]], [eager[-1]])
    q, s, d, i = res[0]                            # tuple unpacking of an entry
    assert q.dtype == torch.int64 and s.dtype == torch.float32 and d.dtype == torch.int64 and i.dtype == torch.int64 and q.shape[1] == 5
    import copy
    import pickle
    assert type(pickle.loads(pickle.dumps([None if r is None else tuple(t.cpu() for t in r) for r in res]))) is list
    assert len(copy.copy(res)) == 4 and len(res + [None]) == 5


@pytest.mark.gpu
def test_kept_batch_is_dropped_when_a_step_had_to_be_repeated_on_the_exact_kernels():
    """a value written in place that leaves the split form's range raises the step's range flag: ``forward`` repeats the step on the exact f32
    kernels, which read a gathered COPY of the feature rows -- a kept batch that holds such a copy must not serve the next call"""
    cfg, model = _model("CONFIG_V")
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    props = [synthetic_proposal(980 + i, n_trajs=10, feat_width=width, num_enti_cats=cfg["num_enti_cats"], len_range=(10, 30)).to(DEV) for i in range(3)]
    for _ in range(3):
        base = model(props, topk=3)
    kept = model._last_pack["packed"]
    assert kept is not None and kept._feats is None
    with torch.no_grad():
        props[1].features_list[2][0, 5] = 3.0e6                      # outside the f16 pair's range
    got = model(props, topk=3)
    assert model.__dict__.get("_last_pack") is None                  # the repeated step gathered f32 rows: batch not kept
    cfg2, fresh = _model("CONFIG_V")
    _same_results(got, fresh([p for p in props], topk=3))
    with torch.no_grad():
        props[1].features_list[2][0, 5] = 0.25
    again = model(props, topk=3)                                       # reads the callers' tensors again, not a stale copy
    _same_results(again, fresh([p for p in props], topk=3))
