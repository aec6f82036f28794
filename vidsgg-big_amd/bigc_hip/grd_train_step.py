"""Host side of the grounding scorer's training step on the HIP path (reference: ``DEBUG._forward_train``
models/grd_model_v5.py:375-389, driven by the loop body of tools/train_vidor.py:596-611).

Split of work:
  * host: the integer bookkeeping of ``prepare_gt_data`` (:253-306: unique query tags, random negative predicates drawn
    with ``torch.randperm`` from the global RNG exactly as the reference does), ``get_gt_labels``' bin ids (:224-250)
    and ``map2bins`` (:391-442) reduced to index lists;
  * device (HIP library): the network forward with saved activations, the labels + five loss terms + their
    gradients, and the whole backward.
Exposed as a ``torch.autograd.Function`` so that ``total_loss.backward()`` / ``optimizer.step()`` work unchanged.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .train_step import grad_scale

LOSS_KEYS = ("pos_cls", "neg_cls", "pos_ct", "neg_ct", "regr")        # order of the reference's loss_dict (:520-526)


def _unique_rows(x):
    """sorted unique rows + for each the ascending indices of its occurrences (utils/utils_func.py:330-345)"""
    uniq, inverse = torch.unique(x, dim=0, return_inverse=True)
    return uniq, [torch.nonzero(inverse == i, as_tuple=True)[0] for i in range(uniq.shape[0])]


def gt_queries(gt_graph, train, n_pred_cats=51):
    """Query tags of one video's ground truth (:253-306).
    -> quint (n,5) int64 [pred, s_cat, o_cat, 0, 0], dura (n,2) int64, target (n_gt_pred,2) f32 normalised spans,
       index_map (list over the U unique tags of GT-predicate indices); n = 2U when ``train`` (negatives appended)."""
    adj = gt_graph.adj_matrix.detach().cpu()
    cats_all = gt_graph.traj_cat_ids.detach().cpu().long()
    dur_all = gt_graph.traj_durations.detach().cpu().long()
    pred_cats = gt_graph.pred_cat_ids.detach().cpu().long()
    so = torch.argmax(adj, dim=-1).t()                                                # (n_pred, 2) subject / object trajectory
    sd, od = dur_all[so[:, 0]], dur_all[so[:, 1]]
    inter = torch.stack([torch.maximum(sd[:, 0], od[:, 0]), torch.minimum(sd[:, 1], od[:, 1])], -1)
    tags = torch.cat([pred_cats[:, None], cats_all[so], inter], -1)                   # [pred, s_cat, o_cat, start, end]
    uniq, index_map = _unique_rows(tags)
    pred_ids, cats, spans = uniq[:, 0], uniq[:, 1:3], uniq[:, 3:]
    if train:
        _, groups = _unique_rows(uniq[:, 1:])
        every = torch.arange(n_pred_cats)
        neg = []
        for members in groups:                      # one draw per (s_cat, o_cat, span) group, in sorted order (:285-303)
            used = torch.zeros(n_pred_cats, dtype=torch.bool)
            used[tags[members, 0]] = True           # as the reference: indices of unique rows applied to the raw tag list
            free = every[~used]
            neg.append(free[torch.randperm(len(free))][:len(members)])
        pred_ids = torch.cat([pred_ids, torch.cat(neg)])
        cats, spans = torch.cat([cats, cats], 0), torch.cat([spans, spans], 0)
    quint = torch.cat([pred_ids[:, None], cats, torch.zeros_like(cats)], -1)
    target = gt_graph.pred_durations.detach().cpu().float() / gt_graph.video_len
    return quint, spans, target, index_map


def bin_entries(target, index_map, n_queries, num_bins, n_clips):
    """``get_gt_labels``' bin ids (:224-250) and ``map2bins`` (:391-442) as index lists.
    -> pos (n_pos,3) [query, bin, gt predicate], neg (n_neg,2) [query, bin], number of (positive entry, clip) pairs with
       centerness > 0 (the mask of :497)."""
    bins = torch.linspace(0, 1, num_bins + 1)
    center = target.mean(dim=-1)
    bin_ids = (center[:, None] - bins[None, :] > 0).sum(dim=-1) - 1
    pos, neg = [], []
    for i, members in enumerate(index_map):
        b = bin_ids[members]
        used = torch.zeros(num_bins, dtype=torch.bool)
        used[b] = True
        pos += [(i, int(bb), int(j)) for bb, j in zip(b.tolist(), members.tolist())]
        neg += [(i, int(bb)) for bb in torch.nonzero(~used, as_tuple=True)[0].tolist()]
    for i in range(len(index_map), n_queries):                      # sampled negatives: every bin is a negative (:436-441)
        neg += [(i, bb) for bb in range(num_bins)]
    clip = torch.linspace(0, 1, n_clips)
    inside = ((clip[None, :] - target[:, 0, None]) > 0) & ((target[:, 1, None] - clip[None, :]) > 0)
    per_gt = inside.sum(dim=1)
    n_mask = int(sum(int(per_gt[j]) for _, _, j in pos))
    return np.asarray(pos, dtype=np.int32).reshape(-1, 3), np.asarray(neg, dtype=np.int32).reshape(-1, 2), n_mask


class _Video(object):
    pass


def _side_streams(engine, dev):
    """streams of the engine for the videos of a training batch (BIGC_GRD_TRAIN_STREAMS, default 8 = the reference's batch size; 0 or 1 = the caller's stream only)"""
    import os
    n = int(os.environ.get("BIGC_GRD_TRAIN_STREAMS", "8"))
    if n <= 1:
        return []
    st = getattr(engine, "_train_streams", None)
    if st is None or len(st) != n or st[0].device != dev:
        st = engine._train_streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return st


def forward(model, engine, video_feature_list, gt_graph_list):
    """-> (loss terms (5,) device tensor in LOSS_KEYS order, context for ``backward``)"""
    lib, c = engine.lib, engine.c
    k, d = c.num_bins, c.dim_hidden
    ldc, ldr = (k + 3) // 4 * 4, (2 * k + 3) // 4 * 4
    stream = _lib.current_stream()
    vids = []
    # dropout follows the module's training flag (models/grd_model_v5.py:125,130,136 use training=self.training); one seed per
    # video, drawn from the global torch RNG right after that video's negative predicates
    drop_p = float(getattr(model, "dropout_p", 0.1)) if model.training else 0.0
    for feat, g in zip(video_feature_list, gt_graph_list):
        assert feat.is_cuda and feat.dtype == torch.float32
        dev = feat.device
        v = _Video()
        v.feat = feat.detach().contiguous()
        v.T = int(feat.shape[0])
        quint, dura, target, index_map = gt_queries(g, True, c.n_pred_emb)
        v.n = int(quint.shape[0])
        pos, neg, v.n_mask = bin_entries(target, index_map, v.n, k, v.T)
        v.n_pos, v.n_neg = len(pos), len(neg)
        # (host arrays now, ONE upload of all the videos' integer tables and one of their float targets behind the loop: six small
        # pageable copies per video each cost the host ~25 us and wait for the stream)
        v.host_i = [quint.to(torch.int32).numpy().reshape(-1), dura.to(torch.int32).numpy().reshape(-1),
                    np.ascontiguousarray(pos.T).reshape(-1), np.ascontiguousarray(neg.T).reshape(-1)]
        v.host_f = target.contiguous().numpy().reshape(-1)
        v.tshape = tuple(target.shape)
        v.video_len = float(g.video_len)
        v.regrs = torch.empty((v.n * v.T, 2 * k), dtype=torch.float32, device=dev)
        v.conf = torch.empty((v.n * v.T, k), dtype=torch.float32, device=dev)
        v.cls = torch.empty((v.n * v.T, k), dtype=torch.float32, device=dev)
        need = lib.grd_train_workspace_bytes(engine.handle, v.T, v.n)
        v.ws = torch.empty((need,), dtype=torch.uint8, device=dev)             # activations live until the backward
        v.pos_v, v.pos_q, v.clip = engine.tables(v.T, dev)
        v.drop = (drop_p, int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if drop_p > 0 else 0)
        vids.append(v)
    # ---- the batch's tables in two uploads, the seed-gradient buffers in one allocation (pieces padded to 16 bytes)
    dev = vids[0].feat.device
    pad4 = lambda n: (n + 3) // 4 * 4
    i_sizes = [[pad4(a.size) for a in v.host_i] for v in vids]
    f_sizes = [pad4(v.host_f.size) for v in vids]
    blob_i = np.zeros((sum(sum(x) for x in i_sizes),), dtype=np.int32)
    blob_f = np.zeros((max(1, sum(f_sizes)),), dtype=np.float32)
    oi = of = 0
    spans = []
    for v, isz, fsz in zip(vids, i_sizes, f_sizes):
        sp = []
        for a, z in zip(v.host_i, isz):
            blob_i[oi:oi + a.size] = a
            sp.append((oi, a.size))
            oi += z
        blob_f[of:of + v.host_f.size] = v.host_f
        spans.append((sp, (of, v.host_f.size)))
        of += fsz
    dev_i, dev_f = _lib.upload(blob_i, dev), _lib.upload(blob_f, dev)
    g_sizes = [(v.n * v.T * ldr, v.n * v.T * ldc, v.n * v.T * ldc) for v in vids]
    seeds = torch.zeros((sum(sum(x) for x in g_sizes),), dtype=torch.float32, device=dev)
    og = 0
    for v, (sp, (fo, fn)), gs_ in zip(vids, spans, g_sizes):
        cut = lambda k: dev_i[sp[k][0]:sp[k][0] + sp[k][1]]
        v.quint, v.dura = cut(0).view(v.n, 5), cut(1).view(v.n, 2)
        v.pos, v.neg = cut(2).view(3, v.n_pos), cut(3).view(2, v.n_neg)
        v.target = dev_f[fo:fo + fn].view(v.tshape)
        v.dregr = seeds[og:og + gs_[0]].view(v.n * v.T, ldr); og += gs_[0]
        v.dconf = seeds[og:og + gs_[1]].view(v.n * v.T, ldc); og += gs_[1]
        v.dcls = seeds[og:og + gs_[2]].view(v.n * v.T, ldc); og += gs_[2]
        v.host_i = v.host_f = None
    ctx_keep = (dev_i, dev_f, seeds)

    def launch(v):
        _lib.check(lib.grd_set_dropout(engine.handle, v.drop[0], v.drop[1]))
        _lib.check(lib.grd_train_forward(engine.handle, _lib.current_stream(), v.feat.data_ptr(), v.T, v.quint.data_ptr(), v.dura.data_ptr(), v.n,
                                         v.video_len, v.pos_v.data_ptr(), v.pos_q.data_ptr(), v.ws.data_ptr(), v.ws.numel(),
                                         v.regrs.data_ptr(), v.conf.data_ptr(), v.cls.data_ptr()))

    # The videos of a batch are independent until the loss (the reference runs them one after the other, models/grd_model_v5.py:377) and
    # each is a few hundred launches that fill a fraction of the chip: from the second step on -- the first records which weights the step
    # prepares (split forms, transposed copies) -- video 0 runs on the caller's stream, where that preparation is launched, and the others
    # on a few streams of the engine's, forked behind it and joined before the loss.  Every tensor is allocated on the caller's stream.
    side = _side_streams(engine, vids[0].feat.device) if (len(vids) > 1 and getattr(engine, "_train_steps", 0) >= 1) else []
    launch(vids[0])
    if side:
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        for i, v in enumerate(vids[1:]):
            st = side[i % len(side)]
            if i < len(side):
                st.wait_event(ready)
            with torch.cuda.stream(st):
                launch(v)
        for st in side[:len(vids) - 1]:
            main.wait_stream(st)
    else:
        for v in vids[1:]:
            launch(v)
    ctx = _Video()
    ctx.vids = vids
    ctx.keep = ctx_keep
    ctx.count_pos = float(sum(v.n_pos * v.T for v in vids))
    ctx.count_neg = float(sum(v.n_neg * v.T for v in vids))
    ctx.count_mask = float(sum(v.n_mask for v in vids))
    ctx.loss = torch.zeros((5,), dtype=torch.float32, device=vids[0].feat.device)
    loss_grad(model, engine, ctx, [1.0] * 5)
    return ctx.loss.clone(), ctx


def loss_grad(model, engine, ctx, gs):
    f = model.loss_factor
    gs_arr = (ctypes.c_float * 5)(*[float(x) for x in gs])
    ctx.loss.zero_()
    ctx.keep[2].zero_()                      # every video's dregr / dconf / dcls (one buffer)
    for v in ctx.vids:
        _lib.check(engine.lib.grd_loss_grad(
            _lib.current_stream(), v.regrs.data_ptr(), v.conf.data_ptr(), v.cls.data_ptr(), v.clip.data_ptr(), v.target.data_ptr(),
            v.pos[0].data_ptr(), v.pos[1].data_ptr(), v.pos[2].data_ptr(), v.n_pos, v.neg[0].data_ptr(), v.neg[1].data_ptr(), v.n_neg,
            v.T, engine.c.num_bins, ctx.count_pos, ctx.count_neg, ctx.count_mask, float(f["classification"]), float(f["centerness"]),
            float(f["regression"]), gs_arr, ctx.loss.data_ptr(), v.dregr.data_ptr(), v.dconf.data_ptr(), v.dcls.data_ptr()))


def backward(model, engine, ctx, grad_terms, named_params):
    """HIP backward of every video; returns one gradient per entry of ``named_params`` (views of one flat buffer)."""
    loss_grad(model, engine, ctx, grad_terms.detach().cpu().tolist())
    # power-of-two scaling of the seed gradients for the split-operand backward GEMMs (see train_step.backward)
    seeds = ctx.keep[2]                      # every video's seed gradients, one buffer
    scale = grad_scale(seeds)
    if scale != 1.0:
        seeds.mul_(scale)
    dev = ctx.vids[0].feat.device
    trainable = [(k, p) for k, p in named_params if p.requires_grad]
    sizes = [(p.numel() + 3) // 4 * 4 for _, p in trainable]
    # (as in the forward: not in a model's first step, whose backward is where the transposed weight copies are made and recorded)
    side = _side_streams(engine, dev) if (len(ctx.vids) > 1 and getattr(engine, "_train_steps", 0) >= 1) else []
    n_slot = max(1, min(len(side), len(ctx.vids)))
    # one flat gradient buffer per stream: the videos of a stream add into it in order, the buffers are added in order at the end
    flats = [torch.zeros((sum(sizes),), dtype=torch.float32, device=dev) for _ in range(n_slot)]
    for k, p in named_params:
        if not p.requires_grad:
            _lib.check(engine.lib.grd_set_grad(engine.handle, k.encode(), None))

    def run(slot):
        off = 0
        for (k, p), sz in zip(trainable, sizes):
            _lib.check(engine.lib.grd_set_grad(engine.handle, k.encode(), flats[slot][off:off + p.numel()].data_ptr()))
            off += sz
        for v in ctx.vids[slot::n_slot]:
            _lib.check(engine.lib.grd_set_dropout(engine.handle, v.drop[0], v.drop[1]))       # the masks of this video's forward
            _lib.check(engine.lib.grd_train_backward(engine.handle, _lib.current_stream(), v.feat.data_ptr(), v.T, v.quint.data_ptr(), v.n,
                                                     v.ws.data_ptr(), v.ws.numel(), v.dregr.data_ptr(), v.dconf.data_ptr(), v.dcls.data_ptr()))

    if side:
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        for slot in range(n_slot):
            side[slot].wait_event(ready)
            with torch.cuda.stream(side[slot]):
                run(slot)
        for slot in range(n_slot):
            main.wait_stream(side[slot])
    else:
        run(0)
    flat = flats[0]
    for other in flats[1:]:
        flat.add_(other)
    views, off = {}, 0
    for (k, p), sz in zip(trainable, sizes):
        views[k] = flat[off:off + p.numel()].view_as(p)
        off += sz
    if scale != 1.0:
        flat.mul_(1.0 / scale)
    engine._train_steps = getattr(engine, "_train_steps", 0) + 1
    engine.last_flat_grad = flat
    return [views.get(k) for k, _ in named_params]


class HipGrdTrainStep(torch.autograd.Function):
    """loss_terms = HipGrdTrainStep.apply(model, video_feature_list, gt_graph_list, *parameters)"""

    @staticmethod
    def forward(ctx, model, video_feature_list, gt_graph_list, *params):
        engine = model.engine(force=True)      # the weights changed since the last step, whether or not a version counter says so (fused optimizers)
        with torch.no_grad():
            terms, tctx = forward(model, engine, video_feature_list, gt_graph_list)
        ctx.model, ctx.engine, ctx.tctx = model, engine, tctx
        return terms

    @staticmethod
    def backward(ctx, grad_terms):
        model = ctx.model
        named = list(model.named_parameters())
        with torch.no_grad():
            grads = backward(model, ctx.engine, ctx.tctx, grad_terms, named)
        ctx.engine.weights_will_change()           # an optimizer step follows (a fused one does not bump the parameters' versions)
        return (None, None, None) + tuple(grads)
