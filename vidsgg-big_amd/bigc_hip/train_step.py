"""Host side of the BIG-C training step on the HIP path (reference: ``BIG_C._forward_train``,
models/model_0v10.py:510-556, and the loop body of tools/train_vidvrd.py:151-164).

Split of work, as in the reference:
  * device (HIP library): forward with saved activations, matching cost, loss terms, the whole backward;
  * host: thresholding of the (device-computed) vIoU tables into the aligned GT adjacency (:583-602) and the Hungarian
    assignment with ``scipy.optimize.linear_sum_assignment`` on a copy of the cost matrix (:637).
The result is exposed as a ``torch.autograd.Function`` so that ``loss.backward()``, ``clip_grad_norm_`` and
``optimizer.step()`` of the reference's training loop work unchanged.
"""
import ctypes
import math
import os
import time

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import _lib
from .packed import PackedBatch

# Host timeline of the step (BIGC_TRAIN_TRACE=1): seconds spent up to each mark since the previous one, summed over steps.  The
# marks add no device synchronisation, so a mark right after a blocking copy shows how long the host waited for the GPU there.
TRACE = {}
_trace_on = os.environ.get("BIGC_TRAIN_TRACE", "0") == "1"
_trace_last = [0.0]


def mark(name):
    if _trace_on:
        now = time.perf_counter()
        if name is not None:
            TRACE[name] = TRACE.get(name, 0.0) + now - _trace_last[0]
        _trace_last[0] = now


def adj_from_viou(vi, adj, viou_th):
    """GT adjacency re-indexed by proposals from the vIoU table ``vi`` (n_proposals, n_gt_traj):
    (2, n_gt_pred, n_proposals) f32 (reference: the tail of ``enti_viou_align`` models/model_0v10.py:583-602).
    A proposal inherits the adjacency column of its best-overlapping GT trajectory when it passes the vIoU
    threshold for some GT trajectory, or when it is the best proposal of a GT trajectory nobody else covers."""
    mask = vi > viou_th
    best = vi.argmax(dim=0)
    none = mask.sum(dim=0) == 0
    mask[best[none], none] = True
    has = mask.any(dim=1)
    pick = vi.argmax(dim=1)
    return adj[:, :, pick] * has[None, None, :].to(adj.dtype)


def viou_tables_launch(lib, packed, gt_graph_list, download=True):
    """Queues the vIoU kernel (``bigc_viou_matrix``: every proposal against every GT trajectory of its video; the reference loops
    over pairs in Python, models/model_0v10.py:576-581, tools/train_vidor.py:110-121) and the copy of its result into pinned host
    memory on the current stream, and returns without waiting: the tables depend on the boxes only, so the step queues them BEFORE
    the model forward and thresholds them on the host while the forward runs."""
    dev = packed.device
    n_traj = [len(g.traj_bboxes) for g in gt_graph_list]
    lens = [int(b.shape[0]) for g in gt_graph_list for b in g.traj_bboxes]
    gt_boxes = _lib.upload(torch.cat([b.detach().float().cpu() for g in gt_graph_list for b in g.traj_bboxes], 0).contiguous(), dev)
    gt_row_off = _lib.upload(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), dev)
    gt_dura = _lib.upload(torch.cat([g.traj_durations.detach().cpu() for g in gt_graph_list], 0).to(torch.int32).contiguous(), dev)
    gt_traj_off = _lib.upload(np.concatenate([[0], np.cumsum(n_traj)]).astype(np.int32), dev)
    g_max = int(max(n_traj))
    out = torch.zeros((packed.n_vid, packed.max_trk, g_max), dtype=torch.float32, device=dev)
    _lib.check(lib.bigc_viou_matrix(_lib.current_stream(), packed.boxes.data_ptr(), packed.trk_row_off.data_ptr(), packed.trk_dura.data_ptr(),
                                    packed.vid_trk_off.data_ptr(), gt_boxes.data_ptr(), gt_row_off.data_ptr(), gt_dura.data_ptr(),
                                    gt_traj_off.data_ptr(), packed.n_vid, packed.max_trk, g_max, out.data_ptr()))
    if not download:                                     # the tables stay on the device (``bigc_align_adjacency`` thresholds them there)
        return None, None, n_traj, (out, gt_boxes, gt_row_off, gt_dura, gt_traj_off)
    host = _lib.staging(dev).download_buffer(out.shape, torch.float32, "viou")
    host.copy_(out, non_blocking=True)
    done = torch.cuda.Event()
    done.record(torch.cuda.current_stream(dev))
    return host, done, n_traj, (out, gt_boxes, gt_row_off, gt_dura, gt_traj_off)


def viou_tables_wait(packed, pending):
    """-> list of (n_proposals, n_gt_traj) f32 host tensors"""
    host, done, n_traj, _keep = pending
    done.synchronize()                                  # waits for the vIoU kernel + copy only, not for what was queued after them
    n_trk = np.diff(packed.vid_trk_off_host)
    return [host[v, :int(n_trk[v]), :n_traj[v]].clone() for v in range(packed.n_vid)]


def viou_tables(lib, packed, gt_graph_list):
    return viou_tables_wait(packed, viou_tables_launch(lib, packed, gt_graph_list))


def align_launch(lib, packed, gt_graph_list, download=True):
    """First half of ``enti_viou_align`` for every video of the batch: reproduces the reference's in-place
    ``gt_graph.traj_durations[:, 1] -= 1`` (:567) and queues the vIoU tables."""
    for g in gt_graph_list:
        g.traj_durations[:, 1] -= 1
    return viou_tables_launch(lib, packed, gt_graph_list, download)


def align_finish(packed, gt_graph_list, pending, viou_th):
    """Second half: the thresholding of the tables into the aligned adjacency, on the host."""
    return [adj_from_viou(vi, g.adj_matrix.detach().cpu().float(), viou_th) for vi, g in zip(viou_tables_wait(packed, pending), gt_graph_list)]


def align_batch(lib, packed, gt_graph_list, viou_th):
    return align_finish(packed, gt_graph_list, align_launch(lib, packed, gt_graph_list), viou_th)


class TrainContext(object):
    """Everything one training step keeps between its forward and its backward."""
    pass


class MatchIndices(object):
    """``model.last_match_indices`` of a step matched on the device: the list of ``(query indices, GT indices)`` per video that
    ``scipy.optimize.linear_sum_assignment`` returns (query indices ascending), downloaded when somebody looks at it."""

    def __init__(self, match_g, n_vid, n_query, check):
        self._m, self._shape, self._check, self._list = match_g, (n_vid, n_query), check, None

    def _get(self):
        if self._list is None:
            m = self._m.cpu().numpy().reshape(self._shape)
            self._check(wait=True)
            self._list = [(np.nonzero(r >= 0)[0], r[r >= 0].astype(np.int64)) for r in m]
            self._m = None
        return self._list

    def __len__(self):
        return self._shape[0]

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, i):
        return self._get()[i]


def match_on_device(gt_graph_list, n_query, max_trk=0):
    """The device matching (``bigc_align_adjacency`` + ``bigc_hungarian``) covers what training batches look like: every video
    has GT trajectories and fewer GT predicates than the model has queries (then every GT predicate is matched and the loss
    normalisers are known without looking at the assignment).  Anything else, or BIGC_MATCH_ON_HOST=1, takes the host route --
    including what the kernels themselves refuse: ``bigc_align_adjacency`` keeps a video's per-proposal arg-max table in
    LDS (at most 8192 proposals), ``bigc_hungarian`` spreads the queries over one workgroup's lanes (at most 1024 queries)."""
    if os.environ.get("BIGC_MATCH_ON_HOST", "0") == "1":
        return False
    if n_query > 1024 or max_trk > 8192:
        return False
    return all(0 < int(g.adj_matrix.shape[1]) < n_query and len(g.traj_bboxes) > 0 for g in gt_graph_list)


def forward(model, engine, packed, gt_graph_list):
    """Device forward + matching. Returns (loss_terms (3,) device tensor, ctx)."""
    lib, c = engine.lib, engine.c
    dev = packed.device
    B, NT, Q, P = packed.n_vid, packed.n_trk, c.num_querys, c.num_pred_cats
    ctx = TrainContext()
    # the feature rows: read in place from the callers' per-tracklet tensors when the batch holds no copy yet and the handle's mode allows
    # it (the first visual-feature GEMM, its weight gradient and the time-mean of the extra columns walk an address table, as the eval
    # forward does), else packed into f32 rows (0.7 ms and 1.1 GB per 64-video step)
    in_place = packed._feats is None and packed.feats_hr8 is None and packed.can_read_in_place() and \
        os.environ.get("BIGC_FEAT_IN_PLACE", "1") != "0" and bool(lib.bigc_train_reads_rows(engine.handle))
    if in_place:
        packed.rows_in_place()
    ctx.packed, ctx.batch = packed, packed.as_struct(need_f32=not in_place)
    need = lib.bigc_train_workspace_bytes(engine.handle, ctypes.byref(ctx.batch))
    # the saved activations belong to THIS step: a second forward before this step's backward (gradient accumulation over
    # micro-batches, loss1 + loss2) must not overwrite them; torch's caching allocator recycles the block after the backward
    ctx.ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    ctx.logits = torch.empty((B * Q, P), dtype=torch.float32, device=dev)
    ctx.att = torch.empty((2 * Q * NT,), dtype=torch.float32, device=dev)
    ctx.on_device = match_on_device(gt_graph_list, Q, packed.max_trk)
    stream = _lib.current_stream()
    mark("fwd: buffers")
    engine.check_match_flags()                                  # (an invalid cost matrix of an EARLIER device-matched step raises here)
    pending = align_launch(lib, packed, gt_graph_list, download=not ctx.on_device)      # vIoU tables first: they depend on the boxes only
    mark("fwd: vIoU launch + GT upload")
    n_trk = np.diff(packed.vid_trk_off_host)
    n_gt = [int(g.adj_matrix.shape[1]) for g in gt_graph_list]
    assert min(n_gt) > 0, "every training video needs at least one GT predicate"
    g_off = np.concatenate([[0], np.cumsum(n_gt)])
    adj_off = np.concatenate([[0], np.cumsum([2 * g * int(n) for g, n in zip(n_gt, n_trk)])]).astype(np.int64)
    gt_pred_host = torch.cat([g.pred_cat_ids.detach().cpu() for g in gt_graph_list]).to(torch.int32)
    n_off = (B + 1 + 3) // 4 * 4                        # [g_off | pad | gt_pred] in one copy
    ints = torch.zeros((n_off + gt_pred_host.numel(),), dtype=torch.int32)
    ints[:B + 1] = torch.from_numpy(g_off.astype(np.int32))
    ints[n_off:] = gt_pred_host
    ints = _lib.upload(ints, dev)
    ctx.g_off, ctx.gt_pred = ints[:B + 1], ints[n_off:]
    g_max = int(max(n_gt))
    if ctx.on_device:
        # the GT adjacency as given (2, G_v, T_v) goes up; the thresholding of the vIoU tables into the aligned adjacency runs on the
        # device, queued BEFORE the model forward like the tables themselves
        n_traj = pending[2]
        raw_off = np.concatenate([[0], np.cumsum([2 * g * t for g, t in zip(n_gt, n_traj)])]).astype(np.int64)
        for g, t in zip(gt_graph_list, n_traj):
            assert tuple(g.adj_matrix.shape) == (2, int(g.adj_matrix.shape[1]), t), "adj_matrix must be (2, n GT predicates, n GT trajectories)"
        offs = _lib.upload(np.concatenate([adj_off[:-1], raw_off[:-1]]), dev)
        ctx.adj_off, raw_off_dev = offs[:B], offs[B:]
        raw = _lib.upload(torch.cat([g.adj_matrix.detach().cpu().float().reshape(-1) for g in gt_graph_list]), dev)
        ctx.gt_adj = torch.empty((int(adj_off[-1]),), dtype=torch.float32, device=dev)
        viou_dev, _gt_boxes, _gt_row_off, _gt_dura, gt_traj_off = pending[3]
        _lib.check(lib.bigc_align_adjacency(stream, viou_dev.data_ptr(), int(viou_dev.shape[1]), int(viou_dev.shape[2]), packed.vid_trk_off.data_ptr(),
                                            gt_traj_off.data_ptr(), raw.data_ptr(), raw_off_dev.data_ptr(), ctx.g_off.data_ptr(), ctx.adj_off.data_ptr(),
                                            B, float(model.positive_vIoU_th), ctx.gt_adj.data_ptr()))
        mark("fwd: GT adjacency upload + align launch")
    _lib.check(lib.bigc_train_forward(engine.handle, stream, ctypes.byref(ctx.batch), ctx.ws.data_ptr(), ctx.ws.numel(),
                                      ctx.logits.data_ptr(), ctx.att.data_ptr()))
    mark("fwd: bigc_train_forward launches")
    if not ctx.on_device:
        # ground truth, aligned to the proposals on the host while the forward runs
        adjs = align_finish(packed, gt_graph_list, pending, model.positive_vIoU_th)
        mark("fwd: vIoU wait + thresholds (host)")
        assert [int(a.shape[1]) for a in adjs] == n_gt
        ctx.adj_off = _lib.upload(adj_off[:-1].copy(), dev)
        ctx.gt_adj = _lib.upload(torch.cat([a.reshape(-1) for a in adjs]), dev)
    cost = torch.zeros((B, Q, g_max), dtype=torch.float32, device=dev)
    _lib.check(lib.bigc_match_cost(stream, ctx.logits.data_ptr(), P, ctx.att.data_ptr(), packed.vid_trk_off.data_ptr(), ctx.g_off.data_ptr(),
                                   ctx.gt_pred.data_ptr(), ctx.gt_adj.data_ptr(), ctx.adj_off.data_ptr(), B, Q, g_max,
                                   float(model.cost_factor["classification"]), float(model.cost_factor["adj_matrix"]), cost.data_ptr()))
    mark("fwd: adjacency upload + cost launch")
    if ctx.on_device:
        # scipy's algorithm on the device: the host neither waits for the forward nor sees the assignment.  Every GT predicate is
        # matched (G_v < Q), so the normalisers of the three means are those of the ground truth.
        ctx.target = torch.empty((B * Q,), dtype=torch.int32, device=dev)
        ctx.match_g = torch.empty((B * Q,), dtype=torch.int32, device=dev)
        _lib.check(lib.bigc_hungarian(stream, cost.data_ptr(), ctx.g_off.data_ptr(), ctx.gt_pred.data_ptr(), B, Q, g_max,
                                      ctx.target.data_ptr(), ctx.match_g.data_ptr(), engine.match_flag(dev).data_ptr()))
        engine.queue_match_flag_check(dev)
        ctx.indices = MatchIndices(ctx.match_g, B, Q, engine.check_match_flags)
        n_pos = int((gt_pred_host != 0).sum())
        ctx.n_pos = float(n_pos)
        ctx.n_neg = float(max(1, B * Q - n_pos))
        ctx.n_adj = float(sum(2 * g * int(n) for g, n in zip(n_gt, n_trk)))
        mark("fwd: Hungarian launch (device)")
    else:
        cost_host = _lib.download(cost, "cost").numpy()      # D2H sync, as the reference's cost_all.cpu() (:637)
        mark("fwd: WAIT for the forward (cost.cpu())")
        target = np.zeros((B, Q), dtype=np.int32)
        match_g = np.full((B, Q), -1, dtype=np.int32)
        ctx.indices = []
        n_adj = 0
        for v in range(B):
            iq, ig = linear_sum_assignment(cost_host[v, :, :n_gt[v]])
            ctx.indices.append((iq, ig))
            target[v, iq] = gt_pred_host[g_off[v] + ig].numpy()
            match_g[v, iq] = ig
            n_adj += 2 * len(iq) * int(n_trk[v])
        mark("fwd: Hungarian (host)")
        ctx.n_pos = float((target != 0).sum())
        ctx.n_neg = float(max(1, (target == 0).sum()))
        ctx.n_adj = float(n_adj)
        both = _lib.upload(np.stack([target.reshape(-1), match_g.reshape(-1)]), dev)
        ctx.target, ctx.match_g = both[0], both[1]
    ctx.ldg = (P + 3) // 4 * 4
    ctx.dlogits = ctx.datt = None                        # the forward wants the three loss terms only; the gradients are the backward's
    ctx.loss = torch.zeros((3,), dtype=torch.float32, device=dev)
    loss_grad(model, engine, ctx, 1.0, 1.0, 1.0)
    mark("fwd: targets upload + loss launch")
    return ctx.loss.clone(), ctx


def loss_grad(model, engine, ctx, gs_pos, gs_neg, gs_adj, upstream=None):
    """loss terms + d logits / d att; ``upstream`` = the three upstream gradients as a (3,) f32 DEVICE tensor instead of host floats"""
    c, packed = engine.c, ctx.packed
    dl, da = (ctx.dlogits.data_ptr(), ctx.datt.data_ptr()) if ctx.dlogits is not None else (None, None)
    if upstream is not None:
        _lib.check(engine.lib.bigc_loss_grad_dev(
            _lib.current_stream(), ctx.logits.data_ptr(), c.num_pred_cats, ctx.att.data_ptr(), packed.vid_trk_off.data_ptr(),
            ctx.g_off.data_ptr(), ctx.gt_adj.data_ptr(), ctx.adj_off.data_ptr(), ctx.target.data_ptr(), ctx.match_g.data_ptr(),
            packed.n_vid, c.num_querys, ctx.n_pos, ctx.n_neg, ctx.n_adj, float(model.loss_factor["classification"]),
            float(model.loss_factor["adj_matrix"]), float(model.neg_weight), upstream.data_ptr(),
            ctx.loss.data_ptr(), dl, ctx.ldg, da))
        return
    _lib.check(engine.lib.bigc_loss_grad(
        _lib.current_stream(), ctx.logits.data_ptr(), c.num_pred_cats, ctx.att.data_ptr(), packed.vid_trk_off.data_ptr(),
        ctx.g_off.data_ptr(), ctx.gt_adj.data_ptr(), ctx.adj_off.data_ptr(), ctx.target.data_ptr(), ctx.match_g.data_ptr(),
        packed.n_vid, c.num_querys, ctx.n_pos, ctx.n_neg, ctx.n_adj, float(model.loss_factor["classification"]),
        float(model.loss_factor["adj_matrix"]), float(model.neg_weight), float(gs_pos), float(gs_neg), float(gs_adj),
        ctx.loss.data_ptr(), dl, ctx.ldg, da))


def grad_scale(*seeds, target_log2=3):
    """power of two that brings the largest seed-gradient magnitude to about 2**target_log2 (1.0 when the seeds are all
    zero or not finite)"""
    seeds = [t for t in seeds if t is not None and t.numel()]
    if not seeds or os.environ.get("BIGC_GRAD_SCALE", "1") == "0":
        return 1.0
    amax = float(torch.stack([t.abs().max() for t in seeds]).max())          # one host sync
    if not (amax > 0.0) or not math.isfinite(amax):
        return 1.0
    return 2.0 ** (target_log2 - math.ceil(math.log2(amax)))


def grad_scale_device(*seeds, target_log2=3):
    """``grad_scale`` without the host sync: the power of two as a 0-dim device tensor"""
    seeds = [t for t in seeds if t is not None and t.numel()]
    if not seeds or os.environ.get("BIGC_GRAD_SCALE", "1") == "0":
        return None
    amax = torch.stack([t.abs().max() for t in seeds]).max()
    ok = torch.isfinite(amax) & (amax > 0)
    return torch.where(ok, torch.exp2(target_log2 - torch.ceil(torch.log2(torch.where(ok, amax, torch.ones_like(amax))))), torch.ones_like(amax))


def backward(model, engine, ctx, grad_terms, named_params):
    """Runs the HIP backward; returns one gradient tensor per entry of ``named_params`` (views of a single flat
    buffer -- the message of the data-parallel all-reduce), ``None`` for frozen parameters."""
    mark("autograd glue between forward and backward")
    dev = ctx.packed.device
    ctx.dlogits = torch.zeros((ctx.packed.n_vid * engine.c.num_querys, ctx.ldg), dtype=torch.float32, device=dev)
    ctx.datt = torch.empty_like(ctx.att)
    if grad_terms.is_cuda and os.environ.get("BIGC_UPSTREAM_ON_HOST", "0") != "1":
        up = grad_terms.detach().to(torch.float32).contiguous()           # read by the kernel where it lies: no wait for the forward
        loss_grad(model, engine, ctx, 1.0, 1.0, 1.0, upstream=up)
        mark("bwd: loss gradient launch (upstream on the device)")
    else:
        g = [float(x) for x in grad_terms.detach().cpu().tolist()]
        mark("bwd: WAIT for the upstream gradient (.cpu())")
        loss_grad(model, engine, ctx, g[0], g[1], g[2])
    # The backward GEMMs split their f32 operands into two f16 (csrc/linear_split.hip): exact for magnitudes inside the
    # f16 range, so the seed gradients are brought to a magnitude of ~8 by a power of two (exact) and the parameter
    # gradients scaled back at the end -- the loss scaling of mixed-precision training, here only to keep f32 accuracy.
    # d att = w (a - t) / max(a (1 - a), 1e-12) (csrc/loss.hip, as torch's BCE backward) reaches 1e10 on a saturated attention
    # weight, while what the GEMMs see of it -- after the double-softmax backward, an f32 kernel, multiplied it by ~a (1 - a) -- is
    # of ordinary size.  Taking the scale from the raw d att would push d logits (and with it every weight gradient) under the
    # split format's floor of 2^-35: the magnitude that counts is the one behind the softmax.
    scale = grad_scale_device(ctx.dlogits, ctx.datt * (ctx.att * (1.0 - ctx.att)))      # (a device scalar: the step does not wait for it)
    if scale is not None:
        ctx.dlogits.mul_(scale)
        ctx.datt.mul_(scale)
    trainable = [(k, p) for k, p in named_params if p.requires_grad]
    flat, views = engine.flat_gradients(trainable, ctx.packed.device)
    mark("bwd: seed scaling + flat gradient buffer")
    _lib.check(engine.lib.bigc_train_backward(engine.handle, _lib.current_stream(), ctypes.byref(ctx.batch), ctx.ws.data_ptr(),
                                              ctx.ws.numel(), ctx.dlogits.data_ptr(), ctx.ldg, ctx.datt.data_ptr()))
    mark("bwd: bigc_train_backward launches")
    if scale is not None:
        flat.div_(scale)                                   # a power of two: exact
    return [views.get(k) for k, _ in named_params]


class HipTrainStep(torch.autograd.Function):
    """loss_terms = HipTrainStep.apply(model, packed, gt_graph_list, *parameters)"""

    @staticmethod
    def forward(ctx, model, packed, gt_graph_list, *params):
        engine = model.engine(force=True)      # the weights changed since the last step, whether or not a version counter says so (fused optimizers)
        mark("register parameters + finalize")
        # dropout follows the module's training flag like the reference's nn.Dropout / nn.MultiheadAttention(dropout=0.1)
        # (models/model_0v10.py:320,326); the per-step seed comes from the global torch RNG
        drop_p = float(getattr(model, "dropout_p", 0.1)) if model.training else 0.0
        drop_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if drop_p > 0 else 0
        engine.set_dropout(drop_p, drop_seed)
        with torch.no_grad():
            terms, tctx = forward(model, engine, packed, gt_graph_list)
        tctx.dropout = (drop_p, drop_seed)
        ctx.model, ctx.engine, ctx.tctx = model, engine, tctx
        model.last_match_indices = tctx.indices
        return terms

    @staticmethod
    def backward(ctx, grad_terms):
        model = ctx.model
        named = list(model.named_parameters())
        ctx.engine.set_dropout(*ctx.tctx.dropout)          # the masks of THIS step's forward
        with torch.no_grad():
            grads = backward(model, ctx.engine, ctx.tctx, grad_terms, named)
        ctx.engine.weights_will_change()           # an optimizer step follows (a fused one does not bump the parameters' versions)
        return (None, None, None) + tuple(grads)
