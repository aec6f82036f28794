"""Host side of the BIG-C training step on the HIP path (reference: ``BIG_C._forward_train``,
models/model_0v10.py:510-556, and the loop body of tools/train_vidvrd.py:151-164).

Split of work, as in the reference:
  * device (HIP library): forward with saved activations, matching cost, loss terms, the whole backward;
  * host: GT-to-proposal alignment by vIoU (a Python loop in the reference too, :559-604) and the Hungarian
    assignment with ``scipy.optimize.linear_sum_assignment`` on a copy of the cost matrix (:637).
The result is exposed as a ``torch.autograd.Function`` so that ``loss.backward()``, ``clip_grad_norm_`` and
``optimizer.step()`` of the reference's training loop work unchanged.
"""
import ctypes

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import _lib
from .packed import PackedBatch


def _viou(boxes_p, boxes_g, lo_p, hi_p, lo_g, hi_g):
    """Volume IoU of two trajectories on their temporal intersection, relative closed frame ranges
    (reference: ``vIoU_ts`` utils/utils_func.py:437-471, +1 pixel convention)."""
    area_p = ((boxes_p[:, 2] - boxes_p[:, 0] + 1) * (boxes_p[:, 3] - boxes_p[:, 1] + 1)).sum()
    area_g = ((boxes_g[:, 2] - boxes_g[:, 0] + 1) * (boxes_g[:, 3] - boxes_g[:, 1] + 1)).sum()
    a, b = boxes_p[lo_p:hi_p + 1], boxes_g[lo_g:hi_g + 1]
    lt = torch.maximum(a[:, :2], b[:, :2])
    rb = torch.minimum(a[:, 2:], b[:, 2:])
    wh = (rb - lt + 1).clamp(min=0.0)
    inter = (wh[:, 0] * wh[:, 1]).sum()
    return inter / (area_p + area_g - inter)


def align_gt_to_proposals(proposal, gt_graph, viou_th):
    """GT adjacency re-indexed by proposals, (2, n_gt_pred, n_proposals) f32 on the CPU
    (reference: ``enti_viou_align`` models/model_0v10.py:559-604, including its in-place
    ``gt_graph.traj_durations[:, 1] -= 1`` at :567)."""
    adj = gt_graph.adj_matrix.detach().cpu()
    n_gt_pred = adj.shape[1]
    n = proposal.num_proposals
    gt_graph.traj_durations[:, 1] -= 1
    pd = proposal.traj_durations.cpu()
    gd = gt_graph.traj_durations.cpu()
    pb = [b.detach().cpu().float() for b in proposal.bboxes_list]
    gb = [b.detach().cpu().float() for b in gt_graph.traj_bboxes]
    s = torch.maximum(pd[:, None, 0], gd[None, :, 0])
    e = torch.minimum(pd[:, None, 1], gd[None, :, 1])
    vi = torch.zeros((n, gd.shape[0]), dtype=torch.float32)
    for p, g in (s <= e).nonzero().tolist():
        vi[p, g] = _viou(pb[p], gb[g], int(s[p, g] - pd[p, 0]), int(e[p, g] - pd[p, 0]), int(s[p, g] - gd[g, 0]), int(e[p, g] - gd[g, 0]))
    mask = vi > viou_th
    best = vi.argmax(dim=0)
    none = mask.sum(dim=0) == 0
    mask[best[none], none] = True
    out = torch.zeros((2, n_gt_pred, n), dtype=torch.float32)
    for p in range(n):
        if bool(mask[p].any()):
            out[:, :, p] = adj[:, :, int(vi[p].argmax())]
    return out


class TrainContext(object):
    """Everything one training step keeps between its forward and its backward."""
    pass


def forward(model, engine, packed, gt_graph_list):
    """Device forward + host matching. Returns (loss_terms (3,) device tensor, ctx)."""
    lib, c = engine.lib, engine.c
    dev = packed.device
    B, NT, Q, P = packed.n_vid, packed.n_trk, c.num_querys, c.num_pred_cats
    ctx = TrainContext()
    ctx.packed, ctx.batch = packed, packed.as_struct()
    need = lib.bigc_train_workspace_bytes(engine.handle, ctypes.byref(ctx.batch))
    if engine._train_ws is None or engine._train_ws.numel() < need or engine._train_ws.device != dev:
        engine._train_ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    ctx.ws = engine._train_ws
    ctx.logits = torch.empty((B * Q, P), dtype=torch.float32, device=dev)
    ctx.att = torch.empty((2 * Q * NT,), dtype=torch.float32, device=dev)
    stream = _lib.current_stream()
    _lib.check(lib.bigc_train_forward(engine.handle, stream, ctypes.byref(ctx.batch), ctx.ws.data_ptr(), ctx.ws.numel(),
                                      ctx.logits.data_ptr(), ctx.att.data_ptr()))
    # ground truth, aligned to the proposals on the host
    n_trk = np.diff(packed.vid_trk_off_host)
    adjs = [align_gt_to_proposals(p, g, model.positive_vIoU_th) for p, g in zip(packed.proposals, gt_graph_list)]
    n_gt = [int(a.shape[1]) for a in adjs]
    assert min(n_gt) > 0, "every training video needs at least one GT predicate"
    g_off = np.concatenate([[0], np.cumsum(n_gt)])
    adj_off = np.concatenate([[0], np.cumsum([2 * g * int(n) for g, n in zip(n_gt, n_trk)])]).astype(np.int64)
    ctx.g_off = torch.from_numpy(g_off.astype(np.int32)).to(dev)
    ctx.adj_off = torch.from_numpy(adj_off[:-1].copy()).to(dev)
    ctx.gt_adj = torch.cat([a.reshape(-1) for a in adjs]).to(dev)
    gt_pred_host = torch.cat([g.pred_cat_ids.detach().cpu() for g in gt_graph_list]).to(torch.int32)
    ctx.gt_pred = gt_pred_host.to(dev)
    g_max = int(max(n_gt))
    cost = torch.zeros((B, Q, g_max), dtype=torch.float32, device=dev)
    _lib.check(lib.bigc_match_cost(stream, ctx.logits.data_ptr(), P, ctx.att.data_ptr(), packed.vid_trk_off.data_ptr(), ctx.g_off.data_ptr(),
                                   ctx.gt_pred.data_ptr(), ctx.gt_adj.data_ptr(), ctx.adj_off.data_ptr(), B, Q, g_max,
                                   float(model.cost_factor["classification"]), float(model.cost_factor["adj_matrix"]), cost.data_ptr()))
    cost_host = cost.cpu().numpy()                      # D2H sync, as the reference's cost_all.cpu() (:637)
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
    ctx.n_pos = float((target != 0).sum())
    ctx.n_neg = float(max(1, (target == 0).sum()))
    ctx.n_adj = float(n_adj)
    ctx.target = torch.from_numpy(target.reshape(-1)).to(dev)
    ctx.match_g = torch.from_numpy(match_g.reshape(-1)).to(dev)
    ctx.ldg = (P + 3) // 4 * 4
    ctx.dlogits = torch.zeros((B * Q, ctx.ldg), dtype=torch.float32, device=dev)
    ctx.datt = torch.empty_like(ctx.att)
    ctx.loss = torch.zeros((3,), dtype=torch.float32, device=dev)
    loss_grad(model, engine, ctx, 1.0, 1.0, 1.0)
    return ctx.loss.clone(), ctx


def loss_grad(model, engine, ctx, gs_pos, gs_neg, gs_adj):
    c, packed = engine.c, ctx.packed
    _lib.check(engine.lib.bigc_loss_grad(
        _lib.current_stream(), ctx.logits.data_ptr(), c.num_pred_cats, ctx.att.data_ptr(), packed.vid_trk_off.data_ptr(),
        ctx.g_off.data_ptr(), ctx.gt_adj.data_ptr(), ctx.adj_off.data_ptr(), ctx.target.data_ptr(), ctx.match_g.data_ptr(),
        packed.n_vid, c.num_querys, ctx.n_pos, ctx.n_neg, ctx.n_adj, float(model.loss_factor["classification"]),
        float(model.loss_factor["adj_matrix"]), float(model.neg_weight), float(gs_pos), float(gs_neg), float(gs_adj),
        ctx.loss.data_ptr(), ctx.dlogits.data_ptr(), ctx.ldg, ctx.datt.data_ptr()))


def backward(model, engine, ctx, grad_terms, named_params):
    """Runs the HIP backward; returns one gradient tensor per entry of ``named_params`` (views of a single flat
    buffer -- the message of the data-parallel all-reduce), ``None`` for frozen parameters."""
    g = [float(x) for x in grad_terms.detach().cpu().tolist()]
    loss_grad(model, engine, ctx, g[0], g[1], g[2])
    trainable = [(k, p) for k, p in named_params if p.requires_grad]
    total = sum(p.numel() for _, p in trainable)
    flat = torch.zeros((total,), dtype=torch.float32, device=ctx.packed.device)
    views, off = {}, 0
    for k, p in trainable:
        n = p.numel()
        assert off % 4 == 0 or True
        views[k] = flat[off:off + n].view_as(p)
        _lib.check(engine.lib.bigc_set_grad(engine.handle, k.encode(), views[k].data_ptr()))
        off += n
    _lib.check(engine.lib.bigc_train_backward(engine.handle, _lib.current_stream(), ctypes.byref(ctx.batch), ctx.ws.data_ptr(),
                                              ctx.ws.numel(), ctx.dlogits.data_ptr(), ctx.ldg, ctx.datt.data_ptr()))
    engine.last_flat_grad = flat
    return [views.get(k) for k, _ in named_params]


class HipTrainStep(torch.autograd.Function):
    """loss_terms = HipTrainStep.apply(model, packed, gt_graph_list, *parameters)"""

    @staticmethod
    def forward(ctx, model, packed, gt_graph_list, *params):
        engine = model.engine()
        with torch.no_grad():
            terms, tctx = forward(model, engine, packed, gt_graph_list)
        ctx.model, ctx.engine, ctx.tctx = model, engine, tctx
        model.last_match_indices = tctx.indices
        return terms

    @staticmethod
    def backward(ctx, grad_terms):
        model = ctx.model
        named = list(model.named_parameters())
        with torch.no_grad():
            grads = backward(model, ctx.engine, ctx.tctx, grad_terms, named)
        return (None, None, None) + tuple(grads)
