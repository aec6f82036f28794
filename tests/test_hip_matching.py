"""The matching of the BIG-C training step on the device (csrc/loss.hip: ``bigc_align_adjacency``, ``bigc_hungarian``,
``bigc_loss_grad_dev``; csrc/ingest.hip: ``bigc_copy_mapped``).  Reference: the host half of ``BIG_C._forward_train`` --
``enti_viou_align`` models/model_0v10.py:583-602 and ``linear_sum_assignment`` :637.  Integer / index work: bit-exact against scipy
itself (also on matrices whose optimum is not unique) and against the host route of the same step."""
import os
import zlib

import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

import cases
from bigc_hip import _lib
from bigc_hip import train_step as TS
from oracle import lsap

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def S():
    return _lib.current_stream()


def _cost_batch(rs, kind, B, nq, gmax):
    n_gt = rs.randint(1, gmax + 1, B)
    n_gt[rs.randint(B)] = gmax
    cost = rs.rand(B, nq, gmax).astype(np.float32)
    if kind == "small integers":
        cost = rs.randint(0, 3, (B, nq, gmax)).astype(np.float32)
    elif kind == "constant":
        cost[:] = 1.0
    elif kind == "duplicated predicates":
        for v in range(B):
            src = rs.randint(0, n_gt[v], gmax)
            cost[v] = cost[v][:, src]
    elif kind == "partly infinite":
        cost = rs.randint(0, 4, (B, nq, gmax)).astype(np.float32)
        cost[rs.rand(B, nq, gmax) < 0.3] = np.inf
    elif kind == "wide range":
        cost = (cost * 10.0 ** rs.randint(-6, 6, (B, nq, gmax))).astype(np.float32)
    return cost, n_gt


def _run_hungarian(lib, cost, n_gt, nq, n_cls=7, seed=0):
    B, _, gmax = cost.shape
    g_off = np.concatenate([[0], np.cumsum(n_gt)]).astype(np.int32)
    gt_pred = np.random.RandomState(seed).randint(1, n_cls, int(g_off[-1])).astype(np.int32)
    c, go, gp = torch.from_numpy(cost).to(DEV), torch.from_numpy(g_off).to(DEV), torch.from_numpy(gt_pred).to(DEV)
    target = torch.full((B * nq,), -7, dtype=torch.int32, device=DEV)
    match = torch.full((B * nq,), -7, dtype=torch.int32, device=DEV)
    bad = torch.zeros((1,), dtype=torch.int32, device=DEV)
    _lib.check(lib.bigc_hungarian(S(), c.data_ptr(), go.data_ptr(), gp.data_ptr(), B, nq, gmax, target.data_ptr(), match.data_ptr(), bad.data_ptr()))
    torch.cuda.synchronize()
    return target.cpu().numpy().reshape(B, nq), match.cpu().numpy().reshape(B, nq), int(bad), g_off, gt_pred


@pytest.mark.parametrize("kind", ["random", "small integers", "constant", "duplicated predicates", "partly infinite", "wide range"])
@pytest.mark.parametrize("nq,gmax", [(8, 5), (32, 31), (192, 24), (192, 120), (70, 9)])
def test_hungarian_is_scipys_assignment(kind, nq, gmax):
    """(192, 120): the cost block of a video no longer fits the LDS beside the algorithm's tables -- read from global memory"""
    lib = _lib.load()
    rs = np.random.RandomState(zlib.crc32(("%s %d %d" % (kind, nq, gmax)).encode()))
    B = 9
    cost, n_gt = _cost_batch(rs, kind, B, nq, gmax)
    target, match, bad, g_off, gt_pred = _run_hungarian(lib, cost, n_gt, nq)
    infeasible = 0
    for v in range(B):
        try:
            iq, ig = linear_sum_assignment(cost[v, :, :n_gt[v]])
        except ValueError:
            infeasible += 1
            assert (match[v] == -1).all() and (target[v] == 0).all()
            continue
        want = -np.ones(nq, dtype=np.int64)
        want[iq] = ig
        assert np.array_equal(match[v], want), (kind, v)
        want_t = np.zeros(nq, dtype=np.int64)
        want_t[iq] = gt_pred[g_off[v] + ig]
        assert np.array_equal(target[v], want_t)
        assert np.array_equal(lsap.assign(cost[v, :, :n_gt[v]]), want)          # the CPU restatement of the same kernel
    assert bad == (1 if infeasible else 0)


@pytest.mark.parametrize("poison", [float("nan"), float("-inf")])
def test_hungarian_flags_what_scipy_rejects(poison):
    lib = _lib.load()
    rs = np.random.RandomState(3)
    cost, n_gt = _cost_batch(rs, "random", 4, 32, 6)
    cost[2, 5, 0] = poison
    with pytest.raises(ValueError):
        linear_sum_assignment(cost[2, :, :n_gt[2]])
    target, match, bad, _, _ = _run_hungarian(lib, cost, n_gt, 32)
    assert bad == 1 and (match[2] == -1).all() and (target[2] == 0).all()
    for v in (0, 1, 3):                                   # the other videos of the batch are matched as usual
        iq, ig = linear_sum_assignment(cost[v, :, :n_gt[v]])
        assert np.array_equal(np.nonzero(match[v] >= 0)[0], iq) and np.array_equal(match[v][iq], ig)


def test_hungarian_rejects_more_predicates_than_queries():
    lib = _lib.load()
    t = torch.zeros((64,), dtype=torch.int32, device=DEV)
    with pytest.raises(AssertionError):
        _lib.check(lib.bigc_hungarian(S(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 1, 8, 8, t.data_ptr(), t.data_ptr(), t.data_ptr()))


@pytest.mark.parametrize("seed", range(6))
def test_align_adjacency_is_the_host_thresholding(seed):
    """random vIoU tables with exact zeros, ties between proposals / trajectories and trajectories nobody passes for"""
    lib = _lib.load()
    rs = np.random.RandomState(seed)
    B = 7
    n_trk = rs.randint(1, 70, B)
    n_traj = rs.randint(1, 12, B)
    n_gt = rs.randint(1, 20, B)
    max_trk, t_max = int(n_trk.max()) + rs.randint(0, 3), int(n_traj.max())
    vi = np.zeros((B, max_trk, t_max), dtype=np.float32)
    adjs, want = [], []
    for v in range(B):
        x = rs.rand(n_trk[v], n_traj[v]).astype(np.float32)
        x[rs.rand(*x.shape) < 0.5] = 0.0
        x = np.round(x * 4) / 4 if seed % 2 else x                 # ties
        if n_traj[v] > 1:
            x[:, rs.randint(n_traj[v])] *= 0.3                     # a trajectory below the threshold everywhere
        vi[v, :n_trk[v], :n_traj[v]] = x
        a = (rs.rand(2, n_gt[v], n_traj[v]) < 0.3).astype(np.float32)
        adjs.append(a)
        w = TS.adj_from_viou(torch.from_numpy(x.copy()), torch.from_numpy(a), 0.5).numpy()
        assert np.array_equal(w, lsap.align(x, a, 0.5))
        want.append(w)
    e_off = np.concatenate([[0], np.cumsum(n_trk)]).astype(np.int32)
    t_off = np.concatenate([[0], np.cumsum(n_traj)]).astype(np.int32)
    g_off = np.concatenate([[0], np.cumsum(n_gt)]).astype(np.int32)
    raw_off = np.concatenate([[0], np.cumsum(2 * n_gt * n_traj)]).astype(np.int64)
    adj_off = np.concatenate([[0], np.cumsum(2 * n_gt * n_trk)]).astype(np.int64)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    vid, eo, to, go, ro, ao = d(vi), d(e_off), d(t_off), d(g_off), d(raw_off[:-1]), d(adj_off[:-1])
    raw = d(np.concatenate([a.reshape(-1) for a in adjs]))
    out = torch.full((int(adj_off[-1]),), float("nan"), device=DEV)
    _lib.check(lib.bigc_align_adjacency(S(), vid.data_ptr(), max_trk, t_max, eo.data_ptr(), to.data_ptr(), raw.data_ptr(), ro.data_ptr(), go.data_ptr(),
                                        ao.data_ptr(), B, 0.5, out.data_ptr()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for v in range(B):
        assert np.array_equal(got[adj_off[v]:adj_off[v + 1]].reshape(want[v].shape), want[v]), v


@pytest.mark.parametrize("nbytes,misalign", [(1, 0), (15, 0), (4096, 0), (100001, 0), (3 << 20, 0), (70000, 4)])
def test_copy_mapped_reads_pinned_memory(nbytes, misalign):
    lib = _lib.load()
    src = torch.empty((nbytes + 64,), dtype=torch.uint8).pin_memory()
    src.copy_(torch.from_numpy(np.random.RandomState(nbytes).randint(0, 256, nbytes + 64).astype(np.uint8)))
    dst = torch.zeros((nbytes + 64,), dtype=torch.uint8, device=DEV)
    _lib.check(lib.bigc_copy_mapped(S(), dst.data_ptr() + misalign, src.data_ptr() + misalign, nbytes))
    torch.cuda.synchronize()
    got = dst.cpu()
    assert torch.equal(got[misalign:misalign + nbytes], src[misalign:misalign + nbytes])
    assert int(got[misalign + nbytes:].sum()) == 0 and int(got[:misalign].sum()) == 0


def test_large_uploads_go_through_the_copy_kernel_and_arrive_intact():
    a = np.random.RandomState(0).randn(300, 257).astype(np.float32)            # 308 KB: over the cap of the runtime's asynchronous copies
    t = _lib.upload(a, DEV)
    b = _lib.upload(torch.arange(100000, dtype=torch.int64), DEV)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), a) and torch.equal(b.cpu(), torch.arange(100000, dtype=torch.int64))


def _same_gradients(g0, g1):
    """equal to a rounding: gradients that are sums of float atomics (bias columns, LayerNorm weights, the bias matrix) differ by that much
    between two runs of the SAME route as well (benchmarks/probes/match_route_diff.py: 4 of 97 parameters at the tiny dims, half of
    them at the benchmark's, up to 3e-7 relative)"""
    for k in g0:
        if not torch.equal(g0[k], g1[k]):
            assert float((g0[k] - g1[k]).abs().max()) <= 2e-6 * float(g0[k].abs().max()), k


def _step(model, props, gts, duras):
    for g, d in zip(gts, duras):
        g.traj_durations = d.clone()
    model.zero_grad()
    total, terms = model(props, gts)
    idx = model.last_match_indices
    (total * 1.0).mean().backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    return float(total), {k: float(v) for k, v in terms.items()}, [(np.asarray(a), np.asarray(b)) for a, b in idx], grads


@pytest.mark.parametrize("name", list(cases.TRAIN_CASES))
def test_step_matched_on_the_device_equals_the_step_matched_on_the_host(name, monkeypatch):
    """same kernels either side of the matching: the aligned adjacency, the assignment, the loss terms and every gradient agree
    (the loss terms are sums of float atomics: equal to rounding, not to the bit)"""
    from test_hip_train import build_train_model
    cfg, variant, sd, props, gts = cases.make_train_case(name)
    model = build_train_model(cfg, variant, sd)
    duras = [g.traj_durations.clone() for g in gts]
    assert TS.match_on_device(gts, cfg["num_querys"])
    monkeypatch.setenv("BIGC_MATCH_ON_HOST", "1")
    monkeypatch.setenv("BIGC_UPSTREAM_ON_HOST", "1")
    t0, terms0, idx0, g0 = _step(model, props, gts, duras)
    assert isinstance(idx0, list)
    monkeypatch.setenv("BIGC_MATCH_ON_HOST", "0")
    monkeypatch.setenv("BIGC_UPSTREAM_ON_HOST", "0")
    t1, terms1, idx1, g1 = _step(model, props, gts, duras)
    assert isinstance(model.last_match_indices, TS.MatchIndices)
    assert len(idx0) == len(idx1)
    for (a, b), (c, d) in zip(idx0, idx1):
        assert np.array_equal(a, c) and np.array_equal(b, d)
    assert abs(t0 - t1) <= 2e-6 * abs(t0)
    for k in terms0:
        assert abs(terms0[k] - terms1[k]) <= 2e-6 * max(abs(terms0[k]), 1e-3), k
    assert set(g0) == set(g1)
    _same_gradients(g0, g1)


def test_upstream_gradient_on_the_device_scales_the_three_terms_like_host_floats(monkeypatch):
    """loss = 2 cls_pos + 0.5 cls_neg + 3 adj: the upstream gradients differ per term"""
    from test_hip_train import build_train_model
    cfg, variant, sd, props, gts = cases.make_train_case("bigc_train_tiny")
    model = build_train_model(cfg, variant, sd)
    duras = [g.traj_durations.clone() for g in gts]
    grads = []
    for host in ("1", "0"):
        monkeypatch.setenv("BIGC_UPSTREAM_ON_HOST", host)
        for g, d in zip(gts, duras):
            g.traj_durations = d.clone()
        model.zero_grad()
        _, terms = model(props, gts)
        (2.0 * terms["cls_pos"] + 0.5 * terms["cls_neg"] + 3.0 * terms["adj"]).backward()
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    _same_gradients(grads[0], grads[1])


def test_invalid_costs_of_a_device_matched_step_raise_at_the_next_look(monkeypatch):
    """scipy raises ValueError inside the step; the device route raises when the indices are read or the next step starts"""
    from test_hip_train import build_train_model
    cfg, variant, sd, props, gts = cases.make_train_case("bigc_train_tiny")
    sd = dict(sd)
    k = [k for k in sd if k.startswith("fc_pred2logits") or "logit" in k][0]
    sd[k] = torch.full_like(sd[k], float("nan"))
    model = build_train_model(cfg, variant, sd)
    duras = [g.traj_durations.clone() for g in gts]
    monkeypatch.setenv("BIGC_MATCH_ON_HOST", "1")
    with pytest.raises(ValueError):
        model(props, gts)
    monkeypatch.setenv("BIGC_MATCH_ON_HOST", "0")
    for g, d in zip(gts, duras):
        g.traj_durations = d.clone()
    model(props, gts)
    with pytest.raises(ValueError):
        list(model.last_match_indices)


def test_stamp_marks_progress_in_a_pinned_word():
    lib = _lib.load()
    words = torch.zeros((2,), dtype=torch.int64).pin_memory()
    flag = torch.zeros((1,), dtype=torch.int32, device=DEV)
    a = torch.randn(2048, 2048, device=DEV)
    for _ in range(6):
        a @ a
    _lib.check(lib.bigc_stamp(S(), words.data_ptr(), 7, flag.data_ptr()))
    flag.fill_(1)
    _lib.check(lib.bigc_stamp(S(), words.data_ptr(), 8, flag.data_ptr()))
    _lib.check(lib.bigc_stamp(S(), words.data_ptr(), 9, None))
    w = words.numpy()
    while w[0] < 9:                                        # the host polls; no runtime call
        pass
    assert int(w[0]) == 9 and int(w[1]) == 8               # the flag was down at mark 7, up at mark 8, not looked at by mark 9
    with pytest.raises(AssertionError):
        _lib.check(lib.bigc_stamp(S(), words.data_ptr() + 4, 1, None))


@pytest.mark.parametrize("name", ["bigc_train_tiny", "bigo_train_tiny", "bigc_train_v64"])
def test_training_step_that_reads_the_feature_rows_in_place_equals_the_one_on_packed_rows(name, monkeypatch):
    """fc_feat2enti.0 (models/model_0v10.py:447), its weight gradient and the time-mean of the I3D / classeme columns (:470) through the
    row-address table of the callers' tensors (bigc_train_reads_rows) against the same step on packed f32 rows"""
    from test_hip_train import build_train_model
    cfg, variant, sd, props, gts = cases.make_train_case(name)
    model = build_train_model(cfg, variant, sd)
    props = [p.to(torch.device(DEV)) for p in props]
    duras = [g.traj_durations.clone() for g in gts]
    assert _lib.load().bigc_train_reads_rows(model.engine().handle) == 1
    seen = []
    real = TS.PackedBatch.rows_in_place
    monkeypatch.setattr(TS.PackedBatch, "rows_in_place", lambda self: (seen.append(1), real(self))[1])
    monkeypatch.setenv("BIGC_FEAT_IN_PLACE", "0")
    t0, terms0, idx0, g0 = _step(model, props, gts, duras)
    assert not seen
    monkeypatch.setenv("BIGC_FEAT_IN_PLACE", "1")
    t1, terms1, idx1, g1 = _step(model, props, gts, duras)
    assert seen, "the step did not take the in-place route"
    for (a, b), (c, d) in zip(idx0, idx1):
        assert np.array_equal(a, c) and np.array_equal(b, d)
    assert abs(t0 - t1) <= 2e-6 * abs(t0)
    _same_gradients(g0, g1)
    # exact-GEMM mode: the dense kernels read packed rows -- the step falls back by itself
    eng = model.engine()
    prev = eng.mode
    eng.set_mode(1, prev[1])
    try:
        assert _lib.load().bigc_train_reads_rows(eng.handle) == 0
        del seen[:]
        _step(model, props, gts, duras)
        assert not seen
    finally:
        eng.set_mode(*prev)


def test_device_matched_training_step_never_waits_for_the_gpu(monkeypatch):
    """forward + backward + clip + fused Adam of a device-matched step: no device-to-host copy, no synchronize -- the host can run a step
    ahead (the reference's step downloads the cost matrix for scipy and aligns the GT adjacency on the host, models/model_0v10.py:583-637)"""
    from test_hip_train import build_train_model
    cfg, variant, sd, props, gts = cases.make_train_case("bigc_train_tiny")
    model = build_train_model(cfg, variant, sd)
    props = [p.to(torch.device(DEV)) for p in props]
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    duras = [g.traj_durations.clone() for g in gts]

    def step():
        for g, d in zip(gts, duras):
            g.traj_durations = d.clone()
        opt.zero_grad()
        total, _terms = model(props, gts)
        total.mean().backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
        opt.step()
        return total
    step()                                                  # (first step: weight preparation, allocator warm-up)
    torch.cuda.synchronize()
    waits = []
    real_cpu, real_item, real_tolist = torch.Tensor.cpu, torch.Tensor.item, torch.Tensor.tolist
    monkeypatch.setattr(torch.Tensor, "cpu", lambda self, *a, **k: (waits.append("cpu") if self.is_cuda else None, real_cpu(self, *a, **k))[1])
    monkeypatch.setattr(torch.Tensor, "item", lambda self: (waits.append("item") if self.is_cuda else None, real_item(self))[1])
    monkeypatch.setattr(torch.Tensor, "tolist", lambda self: (waits.append("tolist") if self.is_cuda else None, real_tolist(self))[1])
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: waits.append("synchronize"))
    monkeypatch.setattr(torch.cuda.Event, "synchronize", lambda self: waits.append("event.synchronize"))
    monkeypatch.setattr(torch.cuda.Stream, "synchronize", lambda self: waits.append("stream.synchronize"))
    total = step()
    monkeypatch.undo()
    assert waits == [], waits
    torch.cuda.synchronize()
    assert float(total) == float(total)                     # (a finite loss came out of the queued work)
