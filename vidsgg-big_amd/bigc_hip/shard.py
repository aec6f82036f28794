"""One-process-per-GPU replacement of the reference's single-process ``VidSGG_DataParallel``
(utils/DataParallel.py:45-113): which videos of a batch a rank owns, and how per-video results
come back together.  Videos are independent, so inference needs no data-path collective; the
only communication is the object gather of the (small) result tuples on rank 0.
"""
import torch


def shard_indices(batch_size, rank, world):
    """Indices of the batch items rank ``rank`` processes.

    Same partition as ``VidSGG_DataParallel.scatter`` (utils/DataParallel.py:68-92): contiguous
    chunks of ``batch_size // world`` in collator order, with the reference's special interleave
    for 4 items on 2 devices (``scatter_42`` :95-113: device 0 gets items 0 and 3, device 1 gets 1
    and 2, pairing the longest with the shortest video of a length-sorted batch).  Unlike the
    reference (which asserts divisibility, :76) a remainder is spread over the first ranks.
    """
    assert 0 <= rank < world
    if batch_size == 4 and world == 2:
        return [0, 3] if rank == 0 else [1, 2]
    base, rem = divmod(batch_size, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def balanced_assignment(costs, world):
    """Size-balanced assignment of independent work items (videos) to ``world`` ranks for INFERENCE ("round-robin
    or size-balanced (by sum L) assignment of videos to 8 processes"; the reference's eval loops go through the videos one by one,
    tools/eval_vidvrd.py:117-121, tools/eval_vidor.py:218-256).  Longest-processing-time greedy: items by decreasing cost, each to the
    rank with the smallest load so far (ties: lowest rank; equal costs keep their order) -- deterministic, so every rank computes the
    same assignment without communicating.  -> list of ``world`` index lists, each in increasing index order."""
    assert world >= 1
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += float(costs[i])
    return [sorted(x) for x in out]


def gather_results(local_results, local_indices, batch_size, dist=None, dst=0):
    """Reassemble the per-video results of all ranks in batch order on rank ``dst`` (None elsewhere).
    Tensors are moved to the CPU before they are pickled."""
    def to_cpu(r):
        return None if r is None else tuple(t.cpu() if isinstance(t, torch.Tensor) else t for t in r)

    payload = [(i, to_cpu(r)) for i, r in zip(local_indices, local_results)]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        gathered = [payload]
    else:
        gathered = [None] * dist.get_world_size() if dist.get_rank() == dst else None
        dist.gather_object(payload, gathered, dst=dst)
        if dist.get_rank() != dst:
            return None
    out = [None] * batch_size
    seen = set()
    for part in gathered:
        for i, r in part:
            assert i not in seen
            seen.add(i)
            out[i] = r
    assert len(seen) == batch_size, "every batch item must be owned by exactly one rank"
    return out


def max_over_ranks(seconds, dist=None, device=None):
    """Timing rule of bench.py: the step time of the job is the slowest rank's."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rank_report(dist=None, device=None, seconds=None, steps=1):
    """What the N > 1 lines of bench.py print so that they prove themselves: every rank's (rank, local rank, host, device ordinal, device
    UUID / name, its own step time) gathered on ALL ranks through the job's process group -- RCCL on the GPU boxes, gloo in the CPU tests.
    -> dict(world, ranks_seen, distinct_devices, devices=[...], ms_per_step_min / _max over ranks).  A line whose ``ranks_seen`` or
    ``distinct_devices`` is smaller than ``n_gpus`` did not run one process per GPU."""
    import os
    import socket
    me = dict(rank=int(os.environ.get("RANK", 0)), local_rank=int(os.environ.get("LOCAL_RANK", 0)), host=socket.gethostname(), pid=os.getpid(),
              device=None, uuid=None, name=None, ms_per_step=None if seconds is None else 1e3 * float(seconds) / max(1, steps))
    if device is not None and torch.device(device).type == "cuda":
        idx = torch.device(device).index
        idx = torch.cuda.current_device() if idx is None else idx
        prop = torch.cuda.get_device_properties(idx)
        me.update(device=int(idx), uuid=str(getattr(prop, "uuid", "")), name=prop.name)
    active = dist is not None and dist.is_initialized()
    if active:
        me["rank"] = dist.get_rank()
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, me)
    else:
        everyone = [me]
    everyone = sorted(everyone, key=lambda d: d["rank"])
    devs = {(d["host"], d["uuid"] if d["uuid"] else d["device"]) for d in everyone}
    ms = [d["ms_per_step"] for d in everyone if d["ms_per_step"] is not None]
    return dict(world=len(everyone), ranks_seen=sorted({d["rank"] for d in everyone}), distinct_devices=len(devs),
                backend=(dist.get_backend() if active else None), devices=everyone,
                ms_per_step_min=round(min(ms), 3) if ms else None, ms_per_step_max=round(max(ms), 3) if ms else None)


def time_allreduce(flat, dist=None, repeats=5):
    """The flat-gradient all-reduce ALONE (bench.py --mode train): ``repeats`` all-reduces of ``flat`` (a scratch copy: the gradients
    are left alone) bracketed by device synchronisation -> dict(us, bytes, algorithmic GB/s = bytes / time, bus GB/s = 2 (n-1)/n of
    that: what each rank moves over its links in a ring / direct reduce-scatter + all-gather)."""
    import time
    if flat is None or dist is None or not dist.is_initialized():
        return None
    buf = flat.clone()
    sync = torch.cuda.synchronize if buf.is_cuda else (lambda: None)
    dist.all_reduce(buf)                     # warm-up (communicator set-up)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(repeats):
        dist.all_reduce(buf)
    sync()
    el = (time.perf_counter() - t0) / repeats
    t = torch.tensor([el], dtype=torch.float64, device=buf.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    n = dist.get_world_size()
    nbytes = buf.numel() * buf.element_size()
    return dict(us=round(1e6 * el, 1), bytes=int(nbytes), world=n, algorithmic_gb_per_s=round(nbytes / el / 1e9, 2),
                bus_gb_per_s=round(nbytes / el / 1e9 * 2 * (n - 1) / n, 2), repeats=repeats)


def flat_gradient_view(params):
    """The gradients of ``params`` as ONE flat tensor WITHOUT copying, when they already live back to back in a single
    storage -- which is how the HIP backward hands them out (bigc_hip/train_step.py: every ``.grad`` is a view of one
    flat fp32 buffer, in parameter order).  Returns None when they do not (plain torch gradients)."""
    grads = [p.grad for p in params]
    if not grads:
        return None
    g0 = grads[0]
    st = g0.untyped_storage()
    base, off = g0.data_ptr(), 0
    for g in grads:                      # back to back in memory ...
        if g.data_ptr() != base + 4 * off or g.dtype != torch.float32 or not g.is_contiguous():
            return None
        off += g.numel()
    if grads[-1].untyped_storage().data_ptr() != st.data_ptr() or (g0.storage_offset() + off) * 4 > st.nbytes():
        return None                      # ... and inside ONE allocation
    return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, g0.storage_offset(), (off,))


def zero_gradients(params):
    """``.grad`` = zeros for every trainable parameter: what a rank whose chunk of the batch is EMPTY (batch smaller than the number of
    ranks) contributes to the gradient all-reduce -- it still has to take part in the collective."""
    for p in params:
        if p.requires_grad:
            p.grad = torch.zeros_like(p)


def allreduce_gradients(params, dist=None, average=True, contributing=None):
    """Data-parallel gradient exchange of the training step: ONE all-reduce over a single flat fp32 buffer
    holding every gradient (RCCL over xGMI on the GPU boxes, gloo in the CPU tests), then the mean over ranks.

    Replaces the reference's single-process ``nn.DataParallel`` reduction (utils/DataParallel.py:65-69, loss
    ``.mean()`` over replicas at tools/train_vidvrd.py:155): each rank computes the loss means over its own
    contiguous chunk of the batch (``shard_indices``), so averaging the per-rank gradients reproduces it.
    The HIP backward already writes every gradient into one flat buffer: that buffer is reduced IN PLACE (no
    concatenation, no copy back); gradients that live in separate tensors are gathered into a flat message first.
    ``contributing``: number of ranks with a non-empty chunk (``min(batch_size, world)``) when some ranks ran ``zero_gradients``.
    Returns the flat reduced buffer (27.8 M floats = 111 MB for the exp2 model)."""
    params = [p for p in params if p.grad is not None]
    if not params:
        return None
    active = dist is not None and dist.is_initialized()      # a one-rank group still runs the collective (RCCL smoke on one GPU)
    flat = flat_gradient_view(params)
    in_place = flat is not None
    if not in_place:
        flat = torch.cat([p.grad.reshape(-1) for p in params])
    if active:
        dist.all_reduce(flat)
        # mean over the replicas that HAD a chunk (``contributing``; default: all ranks): the reference's DataParallel only
        # creates as many replicas as scatter() returns chunks (utils/DataParallel.py:65-69) and averages over those
        n = dist.get_world_size() if contributing is None else int(contributing)
        if average and n > 1:
            flat /= n
    if not in_place:
        off = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    allreduce_gradients.last_in_place = in_place
    return flat


allreduce_gradients.last_in_place = None


def clip_grad_norm_(params, max_norm, flat=None):
    """``torch.nn.utils.clip_grad_norm_(params, max_norm, norm_type=2)`` (the reference's tools/train_vidvrd.py:158) on the flat
    gradient buffer when there is one: one norm and one scale over 111 MB instead of a foreach over ~300 tensors.  ``flat``: the
    return value of ``allreduce_gradients`` (skips the layout check).  Returns the total norm (0-dim tensor)."""
    params = [p for p in params if p.grad is not None]
    if flat is None or not allreduce_gradients.last_in_place:
        flat = flat_gradient_view(params)
    if flat is None:
        return torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm, norm_type=2)
    norm = torch.linalg.vector_norm(flat)
    flat.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))
    return norm
