"""World-size-2 gloo test (CPU) of the process-per-GPU sharding that replaces the reference's
DataParallel shim: partition, object gather in batch order, max-over-ranks timing.  The per-video
work is done by the CPU oracle here (the HIP kernels need a GPU); what is under test is the N>1
host path itself."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

import cases
from bigc_hip.shard import allreduce_gradients, balanced_assignment, gather_results, max_over_ranks, shard_indices, zero_gradients


def test_shard_indices_partition():
    for bs in range(1, 20):
        for world in range(1, 9):
            parts = [shard_indices(bs, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(bs))
            if not (bs == 4 and world == 2):
                assert all(p == list(range(p[0], p[0] + len(p))) for p in parts if p)      # contiguous chunks
                assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard_indices(8, 1, 2) == [4, 5, 6, 7]           # utils/DataParallel.py:80-92
    assert shard_indices(4, 0, 2) == [0, 3] and shard_indices(4, 1, 2) == [1, 2]   # scatter_42 :95-113


def test_balanced_assignment_for_inference():
    """size-balanced, deterministic, a partition; every rank computes it without communicating"""
    import numpy as np
    rs = np.random.RandomState(3)
    for n in (0, 1, 5, 37, 200):
        costs = rs.randint(100, 5000, size=n).tolist()
        for world in (1, 2, 3, 8):
            parts = balanced_assignment(costs, world)
            assert len(parts) == world and sorted(i for p in parts for i in p) == list(range(n))
            assert all(p == sorted(p) for p in parts) and parts == balanced_assignment(costs, world)
            if n >= 4 * world:
                loads = [sum(costs[i] for i in p) for p in parts]
                assert max(loads) - min(loads) <= max(costs)                     # LPT: within one item of each other
    assert balanced_assignment([5, 5, 5, 5], 2) == [[0, 2], [1, 3]]              # equal costs: round-robin in index order
    assert balanced_assignment([1, 100, 1, 1], 2) == [[1], [0, 2, 3]]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import bigc_oracle as O
    cfg, variant, sd, props, topk = cases.make_case("bigc_tiny_edges")
    mine = shard_indices(len(props), rank, world)
    local = O.forward_test(sd, cfg, [props[i] for i in mine], topk, variant)
    dist.barrier()
    res = gather_results(local, mine, len(props), dist)
    slow = max_over_ranks(1.0 + rank, dist)
    # gradient exchange of the training step: per-rank gradients -> mean over ranks through one flat buffer
    ps = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2))]
    ps[0].grad = torch.full((3, 4), float(rank + 1))
    ps[1].grad = torch.arange(5, dtype=torch.float32) * (rank + 1)
    flat = allreduce_gradients(ps, dist)                 # ps[2] has no gradient (frozen): skipped
    assert flat.numel() == 17 and ps[2].grad is None
    assert torch.equal(ps[0].grad, torch.full((3, 4), 1.5)) and torch.equal(ps[1].grad, torch.arange(5, dtype=torch.float32) * 1.5)
    assert allreduce_gradients.last_in_place is False
    # the HIP backward hands out views of ONE flat buffer: that buffer is reduced in place, no cat / copy-back
    buf = torch.arange(17, dtype=torch.float32) * (rank + 1)
    ps[0].grad, ps[1].grad = buf[:12].view(3, 4), buf[12:]
    flat2 = allreduce_gradients(ps, dist)
    assert allreduce_gradients.last_in_place is True and flat2.data_ptr() == buf.data_ptr()
    assert torch.equal(buf, torch.arange(17, dtype=torch.float32) * 1.5) and torch.equal(ps[1].grad, buf[12:])
    # a batch smaller than the world: the rank without a chunk contributes zeros and the mean is over the ranks that had one
    # (the reference's DataParallel only makes as many replicas as scatter() gives chunks, utils/DataParallel.py:65-69)
    qs = [torch.nn.Parameter(torch.zeros(2, 2)), torch.nn.Parameter(torch.zeros(3))]
    if rank == 0:
        qs[0].grad, qs[1].grad = torch.full((2, 2), 4.0), torch.full((3,), 2.0)
    else:
        zero_gradients(qs)
    allreduce_gradients(qs, dist, contributing=1)
    assert torch.equal(qs[0].grad, torch.full((2, 2), 4.0)) and torch.equal(qs[1].grad, torch.full((3,), 2.0))
    if rank == 0:
        full = O.forward_test(sd, cfg, props, topk, variant)
        ok = len(res) == len(full)
        for a, b in zip(res, full):
            ok = ok and ((a is None) == (b is None))
            if a is not None:
                ok = ok and all(torch.equal(x, y) for x, y in zip(a, b))
        q.put((ok, slow))
    else:
        assert res is None
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, slow = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok
    assert slow == 2.0


def _worker4(rank, world, port, q):
    """World 4, an uneven batch of 6 videos: ranks 0 and 1 own two videos each, ranks 2 and 3 one (utils/DataParallel.py:80-92 asserts
    divisibility; the remainder goes to the first ranks here).  A second batch of 3 leaves rank 3 without a chunk: it contributes zeros and
    the mean runs over the three ranks that had one (`contributing`), as the reference's DataParallel makes one replica per chunk (:65-69)."""
    import torch.distributed as dist
    from bigc_hip.shard import rank_report, time_allreduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["LOCAL_RANK"] = str(rank), str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    B = 6
    mine = shard_indices(B, rank, world)
    assert mine == [[0, 1], [2, 3], [4], [5]][rank]
    # per-video "gradients": video i contributes (i + 1) everywhere; a rank's gradient is the MEAN over its own chunk (the loss means are per
    # replica, models/model_0v10.py:648-703), the job's gradient the mean over the ranks
    buf = torch.full((10,), sum(i + 1.0 for i in mine) / len(mine))
    ps = [torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(6))]
    ps[0].grad, ps[1].grad = buf[:4], buf[4:]
    flat = allreduce_gradients(ps, dist, contributing=min(B, world))
    want = (1.5 + 3.5 + 5.0 + 6.0) / 4
    assert allreduce_gradients.last_in_place is True and flat.data_ptr() == buf.data_ptr() and torch.allclose(buf, torch.full((10,), want))
    # batch of 3 on 4 ranks: the empty rank takes part with zeros, the mean is over 3
    B2 = 3
    mine2 = shard_indices(B2, rank, world)
    assert mine2 == ([rank] if rank < 3 else [])
    qs = [torch.nn.Parameter(torch.zeros(5))]
    if mine2:
        qs[0].grad = torch.full((5,), float(10 * (rank + 1)))
    else:
        zero_gradients(qs)
    allreduce_gradients(qs, dist, contributing=min(B2, world))
    assert torch.allclose(qs[0].grad, torch.full((5,), 20.0))
    # the self-proving fields of bench.py's N > 1 lines, and the all-reduce timed alone
    rep = rank_report(dist, None, seconds=0.010 * (rank + 1), steps=2)
    tim = time_allreduce(torch.ones(1 << 16), dist, repeats=2)
    res = gather_results([("v%d" % i,) for i in mine], mine, B, dist)
    if rank == 0:
        q.put((rep, tim, res))
    dist.destroy_process_group()


def test_four_rank_gloo_uneven_batch_and_rank_report():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    rep, tim, res = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert rep["world"] == 4 and rep["ranks_seen"] == [0, 1, 2, 3] and rep["backend"] == "gloo" and len(rep["devices"]) == 4
    assert len({d["pid"] for d in rep["devices"]}) == 4                          # four processes
    assert rep["ms_per_step_min"] == 5.0 and rep["ms_per_step_max"] == 20.0      # each rank's own step time, min / max over ranks
    assert tim["world"] == 4 and tim["bytes"] == 4 << 16 and tim["us"] > 0 and abs(tim["bus_gb_per_s"] - 1.5 * tim["algorithmic_gb_per_s"]) < 0.02
    assert res == [("v%d" % i,) for i in range(6)]


def test_rank_report_without_a_group():
    from bigc_hip.shard import rank_report, time_allreduce
    rep = rank_report(None, None, seconds=0.5, steps=10)
    assert rep["world"] == 1 and rep["ranks_seen"] == [0] and rep["distinct_devices"] == 1 and rep["ms_per_step_max"] == 50.0
    assert time_allreduce(torch.ones(4), None) is None
