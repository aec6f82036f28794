"""Training step of the pairwise baseline ``Base_C`` on the HIP path (reference: ``models/model_pairwise_baseline.py:276-312``;
loop body ``tools/train_vidor.py:305-330``).

Host side: concatenation of the per-video ``pairid2trajids`` (shifted to global tracklet rows) and multi-hot labels, the
workspace, and the autograd plumbing.  Front-end, pair head, the BCE-with-logits mean over all labelled pairs of the batch
and the whole backward run in the HIP library (``bigc_pair_train_forward`` / ``bigc_bce_logits_f32`` /
``bigc_pair_train_backward``).  Gradients are views of one flat buffer, as in ``train_step.py``."""
import ctypes

import numpy as np
import torch

from . import _lib
from .train_step import grad_scale


def labels_from_viou(vi, gt_pred_cats, gt_so, viou_th, num_pred_cats):
    """Label assignment of the reference's ``prop_pair_to_gt_pred`` (tools/train_vidor.py:131-158) for one video: every
    ordered proposal pair (s, o), s != o, whose subject / object overlap the subject / object GT trajectory of a GT
    predicate with vIoU > ``viou_th`` gets that predicate.  ``vi`` (n_prop, n_gt_traj), ``gt_so`` (n_gt_pred, 2).
    -> (pairid2trajids (n, 2) int64, multihot (n, P) f32) in the reference's order (first hit: GT predicates in order,
    pairs row-major), or (None, None) when nothing is hit (the reference skips such videos, :246-249, 307-309)."""
    hit = (vi > viou_th).numpy()
    n = hit.shape[0]
    keys, rows = {}, []
    for g in range(len(gt_pred_cats)):
        subs, objs = np.nonzero(hit[:, int(gt_so[g, 0])])[0], np.nonzero(hit[:, int(gt_so[g, 1])])[0]
        for s_ in subs:                                       # row-major over the pairs (s, o), s != o
            for o_ in objs:
                if s_ == o_:
                    continue
                k = int(s_) * n + int(o_)
                if k not in keys:
                    keys[k] = len(rows)
                    rows.append(((int(s_), int(o_)), []))
                rows[keys[k]][1].append(int(gt_pred_cats[g]))
    if not rows:
        return None, None
    ids = torch.tensor([r[0] for r in rows], dtype=torch.int64)
    multihot = torch.zeros((len(rows), num_pred_cats), dtype=torch.float32)
    for i, (_, cats) in enumerate(rows):
        multihot[i, cats] = 1.0
    return ids, multihot


def assign_pair_labels(lib, packed, gt_graph_list, viou_th, num_pred_cats):
    """``prop_pair_to_gt_pred`` for the videos of a PackedBatch: vIoU tables from ``bigc_viou_matrix`` (the reference's
    Python double loop, :110-121), then ``labels_from_viou``.  GT subject / object = argmax of the adjacency (:61)."""
    from .train_step import viou_tables
    out = []
    for vi, g in zip(viou_tables(lib, packed, gt_graph_list), gt_graph_list):
        so = torch.argmax(g.adj_matrix.detach().cpu(), dim=-1).t()
        out.append(labels_from_viou(vi, g.pred_cat_ids.detach().cpu().tolist(), so.numpy(), viou_th, num_pred_cats))
    return out


class PairTrainCtx:
    pass


def forward(model, engine, packed, pos_id_list, label_list):
    """-> (loss terms tensor (1,) = [cls], ctx)."""
    lib, c, dev = engine.lib, engine.c, packed.device
    assert len(pos_id_list) == len(label_list) == packed.n_vid
    off = packed.vid_trk_off_host
    so, labels = [], []
    for v, (ids, lab) in enumerate(zip(pos_id_list, label_list)):
        ids = ids.detach().to("cpu", torch.int64)
        n = int(off[v + 1] - off[v])
        assert ids.dim() == 2 and ids.shape[1] == 2 and lab.shape == (ids.shape[0], c.num_pred_cats), "pairid2trajids (n,2) / multihot (n,P)"
        assert ids.numel() == 0 or (int(ids.min()) >= 0 and int(ids.max()) < n), "pair ids out of range for video %d" % v
        so.append(ids + int(off[v]))
        labels.append(lab.to(dev, torch.float32))
    ctx = PairTrainCtx()
    ctx.packed = packed
    ctx.pair_so = torch.cat(so).to(torch.int32).contiguous().to(dev)
    ctx.labels = torch.cat(labels).contiguous()
    ctx.n_pairs = int(ctx.pair_so.shape[0])
    assert ctx.n_pairs > 0, "Base_C training needs at least one labelled pair in the batch"
    P = c.num_pred_cats
    ctx.batch = packed.as_struct()
    need = lib.bigc_pair_train_workspace_bytes(engine.handle, ctypes.byref(ctx.batch), ctx.n_pairs)
    ctx.ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    ctx.logits = torch.empty((ctx.n_pairs, P), dtype=torch.float32, device=dev)
    _lib.check(lib.bigc_pair_train_forward(engine.handle, _lib.current_stream(), ctypes.byref(ctx.batch), ctx.pair_so.data_ptr(), ctx.n_pairs,
                                           ctx.ws.data_ptr(), ctx.ws.numel(), ctx.logits.data_ptr()))
    ctx.ldg = (P + 3) // 4 * 4
    ctx.dlogits = torch.zeros((ctx.n_pairs, ctx.ldg), dtype=torch.float32, device=dev)
    ctx.loss = torch.zeros((1,), dtype=torch.float32, device=dev)
    _lib.check(lib.bigc_bce_logits_f32(_lib.current_stream(), ctx.logits.data_ptr(), P, ctx.labels.data_ptr(), P, ctx.n_pairs, P, 1.0,
                                       ctx.loss.data_ptr(), ctx.dlogits.data_ptr(), ctx.ldg))
    return ctx.loss.clone(), ctx


def backward(model, engine, ctx, grad_terms, named_params):
    dev = ctx.packed.device
    g = float(grad_terms.detach().reshape(-1)[0])
    seed = ctx.dlogits * g if g != 1.0 else ctx.dlogits
    scale = grad_scale(seed)
    if scale != 1.0:
        seed = seed * scale
    trainable = [(k, p) for k, p in named_params if p.requires_grad]
    flat, views = engine.flat_gradients(trainable, dev)
    _lib.check(engine.lib.bigc_pair_train_backward(engine.handle, _lib.current_stream(), ctypes.byref(ctx.batch), ctx.pair_so.data_ptr(),
                                                   ctx.n_pairs, ctx.ws.data_ptr(), ctx.ws.numel(), seed.data_ptr(), ctx.ldg))
    if scale != 1.0:
        flat.mul_(1.0 / scale)
    return [views.get(k) for k, _ in named_params]


class HipPairTrainStep(torch.autograd.Function):
    """loss_terms (1,) = HipPairTrainStep.apply(model, packed, pos_id_list, label_list, *parameters)"""

    @staticmethod
    def forward(ctx, model, packed, pos_id_list, label_list, *params):
        engine = model.engine(force=True)      # the weights changed since the last step, whether or not a version counter says so (fused optimizers)
        with torch.no_grad():
            terms, tctx = forward(model, engine, packed, pos_id_list, label_list)
        ctx.model, ctx.engine, ctx.tctx = model, engine, tctx
        return terms

    @staticmethod
    def backward(ctx, grad_terms):
        model = ctx.model
        named = list(model.named_parameters())
        with torch.no_grad():
            grads = backward(model, ctx.engine, ctx.tctx, grad_terms, named)
        ctx.engine.weights_will_change()           # an optimizer step follows (a fused one does not bump the parameters' versions)
        return (None, None, None, None) + tuple(grads)
