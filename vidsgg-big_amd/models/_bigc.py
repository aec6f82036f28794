"""Shared host-side body of the two BIG-C drop-in modules.

The module tree below exists to own parameters under the reference's ``state_dict`` keys
(reference constructors models/model_0v10.py:241-336 and
models/model_0v7.py:241-350).  ``nn.Linear`` / ``nn.MultiheadAttention`` / ``nn.LayerNorm``
are used as parameter containers only: ``forward`` never calls them, all arithmetic runs in
the HIP library through ``bigc_hip.engine.BigcEngine``.
"""
import numpy as np
import weakref

import torch
from torch import nn

from bigc_hip.engine import BigcEngine
from bigc_hip.packed import PackedBatch


def _mlp(*dims, final_relu=True):
    """Sequential whose Linear layers sit at indices 0, 2, ... like the reference's."""
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2 or final_relu:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class _EncoderLayerParams(nn.Module):
    """keys: self_attn.*, linear1, linear2, norm1, norm2 (models/model_0v10.py:72-82)"""

    def __init__(self, d, nhead, d_ffn):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, nhead, dropout=0.1)
        self.linear1 = nn.Linear(d, d_ffn)
        self.linear2 = nn.Linear(d_ffn, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)


class _DecoderLayerParams(nn.Module):
    """keys: self_attn.*, fc_rolewise.{0,1}.{0,2}, fc_enti2att, fc_pred2att, fc2.{0,3}, norm{1,2,3}
    (models/model_0v10.py:144-173)"""

    def __init__(self, d_pred, nhead, d_enti, d_att, d_ffn):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_pred, nhead, dropout=0.1)
        self.fc_rolewise = nn.ModuleList([_mlp(d_enti, d_pred, d_pred, final_relu=False) for _ in range(2)])
        self.fc_enti2att = nn.Linear(d_enti, d_att)
        self.fc_pred2att = nn.Linear(d_pred, d_att)
        self.fc2 = nn.Sequential(nn.Linear(d_pred, d_ffn), nn.ReLU(), nn.Dropout(0.1), nn.Linear(d_ffn, d_pred))
        self.norm1 = nn.LayerNorm(d_pred)
        self.norm2 = nn.LayerNorm(d_pred)
        self.norm3 = nn.LayerNorm(d_pred)


class _LazyResults(list):
    """The per-video result list of an eval forward -- a real ``list`` (``len``, indexing, slicing, iteration, ``==`` behave as the
    reference's list of tuples / ``None``) whose entries are made when the list is first READ (all of them then, by one ``split`` per array):
    entry v is ``None`` for a video without triplets, else the tuple of the four views [off_v : off_v + n_v] of the batch's compacted
    arrays.  Cutting 4 x 256 views costs 0.45 ms per call, more than the GPU side of small batches; a caller that reads the results pays it
    at its first read, a caller that only forwards the list (or reads nothing) does not."""
    __slots__ = ("_arrays", "_sizes")

    def __init__(self, arrays, off, sizes):
        super().__init__([None] * len(sizes))
        self._arrays, self._sizes = arrays, sizes

    def _make_all(self):
        """every entry at once, at first read: one ``split`` per array (0.1 ms each for 256 videos; cutting the views one by one from
        Python costs three times that)"""
        if self._arrays is None:
            return
        parts = [a.split(self._sizes) for a in self._arrays]
        for i, n in enumerate(self._sizes):
            if n > 0:
                list.__setitem__(self, i, (parts[0][i], parts[1][i], parts[2][i], parts[3][i]))
        self._arrays = None

    def __getitem__(self, i):
        self._make_all()
        return list.__getitem__(self, i)

    def __iter__(self):
        self._make_all()
        return list.__iter__(self)

    def __reversed__(self):
        self._make_all()
        return list.__reversed__(self)

    def _all(fn):                                           # (whole-list operations see the finished list)
        def wrapped(self, *a, **k):
            self._make_all()
            return fn(self, *a, **k)
        wrapped.__name__ = fn.__name__
        return wrapped

    __eq__ = _all(list.__eq__)
    __ne__ = _all(list.__ne__)
    __repr__ = _all(list.__repr__)
    __add__ = _all(list.__add__)
    __mul__ = _all(list.__mul__)
    __contains__ = _all(list.__contains__)
    __reduce_ex__ = _all(lambda self, proto: (list, (list.__getitem__(self, slice(None)),)))
    copy = _all(list.copy)
    count = _all(list.count)
    index = _all(list.index)
    pop = _all(list.pop)
    remove = _all(list.remove)
    reverse = _all(list.reverse)
    sort = _all(list.sort)
    append = _all(list.append)                              # (mutators too: entries are placed by index when the list is first read)
    extend = _all(list.extend)
    insert = _all(list.insert)
    clear = _all(list.clear)
    __setitem__ = _all(list.__setitem__)
    __delitem__ = _all(list.__delitem__)
    __iadd__ = _all(list.__iadd__)
    __imul__ = _all(list.__imul__)
    del _all
    __hash__ = None


class _Pending(object):
    """an eval forward that has been queued on the GPU (``BigcBase.forward_async``)"""


    def __init__(self, model, proposal_list, topk):
        self.model, self.topk, self.n = model, topk, len(proposal_list)
        self.parts = []                                # (indices into proposal_list, packed batch, outputs, event)
        keep = [i for i, p in enumerate(proposal_list) if p.num_proposals > 0]
        if not keep:
            return
        device = model.pred_query_init.device
        hd = model.config["dim_enti"] // model.config["n_att_head"]
        single_pass = 256 if hd in (8, 16) else 192
        classes = {}
        for i in keep:                                 # a video is only batched with videos of its own size class (see forward)
            n = proposal_list[i].num_proposals
            classes.setdefault(0 if n <= 64 else (1 if n <= single_pass else 2), []).append(i)
        for idx in classes.values():
            packed = PackedBatch([proposal_list[i] for i in idx], device, "lazy")
            out = model.forward_packed(packed, topk)
            open_now = model.__dict__.get("_pending_open", 0)
            assert open_now < 8, "more than eight forwards pending: call .result() on the earlier ones first (their row offsets travel through a ring of eight pinned buffers)"
            object.__setattr__(model, "_pending_open", open_now + 1)
            host = self._pinned_buffer(model, out["_counts"].numel())
            host.copy_(out["_counts"], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            out["_host_pending"] = (lambda ev=ev, host=host: (ev.synchronize(), host.clone())[1])
            self.parts.append((idx, packed, out))

    @staticmethod
    def _pinned_buffer(model, n):
        """the next of the MODEL's eight pinned int32 buffers (allocated once: every pinned allocation maps pages into the GPU's address
        space; per model, so that two models driven from two threads do not share them)"""
        ring = model.__dict__.get("_pending_ring")
        if ring is None or ring[0][0].numel() < n:
            ring = ([torch.empty((max(4096, 2 * n),), dtype=torch.int32).pin_memory() for _ in range(8)], [0])
            object.__setattr__(model, "_pending_ring", ring)
        bufs, nxt = ring
        buf = bufs[nxt[0] % len(bufs)][:n]
        nxt[0] += 1
        return buf

    def __del__(self):
        if getattr(self, "parts", None):                # dropped without .result(): its slot count must not leak
            self.model.__dict__["_pending_open"] = max(0, self.model.__dict__.get("_pending_open", 0) - len(self.parts))

    def result(self):
        res = [None] * self.n
        self.model.__dict__["_pending_open"] = max(0, self.model.__dict__.get("_pending_open", 0) - len(self.parts))
        for idx, packed, out in self.parts:
            try:
                self.model.counts(out)
            except OverflowError:                       # an operand left the split range: this batch once more on the exact kernels
                out = self.model.forward_checked(packed, self.topk)
            for i, r in zip(idx, self.model.unpack_results(len(idx), list(range(len(idx))), out)):
                res[i] = r
        self.parts = []
        return res


class BigcBase(nn.Module):
    variant = None          # "vidvrd" | "vidor"
    default_topk = 10

    def __init__(self, config, is_train=True):
        super().__init__()
        self.is_train = is_train
        self.config = dict(config)
        c = config
        self.num_pred_cats, self.num_enti_cats = c["num_pred_cats"], c["num_enti_cats"]
        self.dim_feat, self.dim_clsme = c["dim_feat"], c["dim_clsme"]
        self.dim_enti, self.dim_pred, self.dim_att, self.dim_ffn = c["dim_enti"], c["dim_pred"], c["dim_att"], c["dim_ffn"]
        self.enco_pool_len = c["enco_pool_len"]
        self.n_enco_layers, self.n_deco_layers = c["n_enco_layers"], c["n_deco_layers"]
        self.n_att_head, self.num_querys = c["n_att_head"], c["num_querys"]
        self.num_anchors = self.num_querys
        self.neg_weight = c["neg_weight"]
        self.loss_factor, self.cost_factor = c["loss_coeff_dict"], c["cost_coeff_dict"]
        self.positive_vIoU_th = c["positive_vIoU_th"]
        self._build_embeddings(c)
        bias = torch.from_numpy(np.load(c["bias_matrix_path"])).float()
        assert bias.shape == (self.num_enti_cats, self.num_enti_cats, self.num_pred_cats)
        self.bias_matrix = nn.Parameter(bias, requires_grad=True)

        de = self.dim_enti
        self.fc_feat2enti = _mlp(self.dim_feat, de, de)
        self.fc_bbox2enti = _mlp(8, de, de)
        self.conv_feat2enti = nn.Conv1d(2 * de, de, kernel_size=3, padding=1, stride=2)
        self.fc_enti2enco = _mlp(de * self.enco_pool_len, de, de)
        self._build_extra_layers(c)
        self.encoder_layers = nn.ModuleList(
            [_EncoderLayerParams(de, self.n_att_head, self.dim_ffn) for _ in range(self.n_enco_layers)])
        self.decoder_layers = nn.ModuleList(
            [_DecoderLayerParams(self.dim_pred, self.n_att_head, de, self.dim_att, self.dim_ffn)
             for _ in range(self.n_deco_layers)])
        self._build_head(c)
        self._reset_parameters()
        self._engine = None
        object.__setattr__(self, "_state_items", None)
        object.__setattr__(self, "_tree_sig", ([], []))

    # ---- hooks for the two variants -----------------------------------------------------
    def _build_embeddings(self, c):
        raise NotImplementedError

    def _build_extra_layers(self, c):
        pass

    def _build_head(self, c):
        raise NotImplementedError

    def _reset_parameters(self):
        raise NotImplementedError

    # ---- forward --------------------------------------------------------------------------
    def forward(self, proposal_list, gt_graph_list=None, topk=None):
        """Same contract as the reference (models/model_0v10.py:359-388): mode is fixed by
        ``is_train``; eval returns one entry per video, ``None`` for an empty video or when no
        subject/object pair overlaps in time, else ``(uniq_quintuples i64 (U,5), uniq_scores f32
        (U,3), uniq_dura_inters i64 (U,2), uniq_query_ids i64 (U,))`` on the input device."""
        if topk is None:
            topk = self.default_topk
        if self.is_train:
            assert gt_graph_list is not None
            return self._forward_train(proposal_list, gt_graph_list)
        self.topk = topk
        if isinstance(proposal_list, PackedBatch):
            return self.unpack_results(proposal_list.n_vid, list(range(proposal_list.n_vid)),
                                       self.forward_checked(proposal_list, topk))
        keep = [i for i, p in enumerate(proposal_list) if p.num_proposals > 0]
        if not keep:
            return [None for _ in proposal_list]
        device = self.pred_query_init.device
        # The library picks its attention kernels per batch from the LARGEST video (fused role attention up to 64 tracklets, scores in
        # registers up to 192; csrc/model.hip, csrc/attention.hip): videos are therefore only batched with videos of their own class,
        # so that a video's result does not depend on its company -- it is, bit for bit, what the video gives alone (the reference
        # scores one video at a time, models/model_0v10.py:376).
        # (class limits follow the kernels' own: 64 tracklets = the fused role attention; then the attention core's single-pass limit,
        # csrc/attention.hip: 192 keys at head_dim 64 / 32, 256 at head_dim 16 / 8.  A PackedBatch given directly is run as ONE batch:
        # whoever packs videos of different classes together gets, for the smaller ones, the larger class's kernels.)
        hd = self.config["dim_enti"] // self.config["n_att_head"]
        single_pass = 256 if hd in (8, 16) else 192
        classes = {}
        for i in keep:
            n = proposal_list[i].num_proposals
            classes.setdefault(0 if n <= 64 else (1 if n <= single_pass else 2), []).append(i)
        if len(classes) == 1 and len(keep) == len(proposal_list):
            # the common case -- one class, no empty video: the results are handed out as a lazy list (see _LazyResults)
            packed = self._pack(proposal_list, device)
            out = self.forward_checked(packed, topk)
            self._remember_pack(proposal_list, packed)
            return self.unpack_results(len(proposal_list), None, out, lazy=True)
        res = [None] * len(proposal_list)
        for idx in classes.values():
            packed = PackedBatch([proposal_list[i] for i in idx], device, "lazy")      # the engine gathers the rows in the form its forward reads
            for i, r in zip(idx, self.unpack_results(len(idx), list(range(len(idx))), self.forward_checked(packed, topk))):
                res[i] = r
        return res

    # ---- the same list of containers again: per-epoch evaluation over a device-resident set, benchmarks ------------------------------
    def _pack(self, proposal_list, device):
        """PackedBatch of the list -- the previous call's when ``proposal_list`` holds the same containers with the same tensors (checked by
        identity, element by element: 0.1 ms for 256 videos where the tables cost 1 ms to build and upload).  Only what the batch copies --
        boxes, categories, scores, durations -- is gathered again (``refresh_values``); the feature rows are read in place anyway.
        Two stages, so that a loop over ever new videos pays one weak reference per call: a call remembers (length, first container); only
        when the NEXT call starts with the same container is that call's batch kept with its full signature; the third call reuses it."""
        c = self.__dict__.get("_last_pack")
        n = len(proposal_list)
        if c is not None and c["packed"] is not None and c["device"] == device and c["n"] == n and all(a() is b for a, b in zip(c["refs"], proposal_list)) \
                and PackedBatch.signature_holds(c["sig"], proposal_list):
            packed = c["packed"]
            packed.proposals = proposal_list
            packed.refresh_values()
            return packed
        repeat = c is not None and c["device"] == device and c["n"] == n and c["first"]() is proposal_list[0]
        object.__setattr__(self, "_last_pack", dict(device=device, n=n, first=weakref.ref(proposal_list[0]), packed=None, keep_next=repeat))
        return PackedBatch(list(proposal_list), device, "lazy")

    def _remember_pack(self, proposal_list, packed):
        """keep the batch for ``_pack`` -- if the list looks like a repeat (see ``_pack``), is made of this package's containers and the forward
        read their rows in place (a batch that holds a COPY of the features would go stale when the caller writes into its tensors).  Holds
        weak references only; the batch's own device memory is its tables and boxes (14 MB for 256 videos)."""
        c = self.__dict__.get("_last_pack")
        if c is None:
            return
        if c["packed"] is packed:
            packed.proposals = None                 # (the kept batch must not keep the containers alive)
            if packed._feats is not None or packed.feats_hr8 is not None:
                # the step was repeated on the exact kernels (range flag): the batch now holds a COPY of the feature rows, which the next
                # call would read instead of the callers' tensors -- not kept
                object.__setattr__(self, "_last_pack", None)
            return
        if not c.get("keep_next"):
            return
        sig = packed.signature() if (packed._feats is None and packed.feats_hr8 is None and packed.feat_row_ptr is not None) else None
        if sig is None:
            return
        packed.proposals = None
        c.update(refs=[weakref.ref(p) for p in proposal_list], sig=sig, packed=packed)

    def forward_async(self, proposal_list, topk=None):
        """``forward`` in two halves, for loops over many batches (tools/eval_*.py): this call packs the batch and QUEUES everything on the
        GPU -- forward, construct_triplet, compaction and a non-blocking copy of the row offsets into pinned memory -- and returns at once;
        ``.result()`` of the returned object waits for that copy and builds the per-video tuples.  The host side of batch i + 1 (tables,
        launches) then runs while the GPU works on batch i:

            pending = model.forward_async(batches[0])
            for nxt in batches[1:]:
                upcoming = model.forward_async(nxt); results = pending.result(); pending = upcoming

        The results are those of ``forward`` (same kernels, same bits; the exact-mode rerun of a step whose range flag is raised happens
        inside ``.result()``).  The feature rows are read IN PLACE while the batch is in flight: the pending object keeps the containers
        alive, but their tensors must not be written before ``.result()`` returns.  At most eight forwards may be pending at a time (the
        ring of pinned buffers the row offsets are copied into)."""
        if topk is None:
            topk = self.default_topk
        assert not self.is_train, "forward_async is an eval entry point"
        return _Pending(self, proposal_list, topk)

    def _forward_train(self, proposal_list, gt_graph_list):
        """Training contract of the reference (models/model_0v10.py:510-556): returns
        ``(total_loss, {"cls_pos", "cls_neg", "adj"})``; ``total_loss.backward()`` runs the HIP backward and
        fills ``.grad`` of every trainable parameter. In ``.train()`` mode (the default, as in the reference's training scripts) the
        transformer layers apply dropout p=0.1 with counter-based masks; ``.eval()`` switches it off (parity tests)."""
        from bigc_hip.train_step import HipTrainStep
        assert len(proposal_list) == len(gt_graph_list)
        assert all(p.num_proposals > 0 for p in proposal_list), "videos without proposals are filtered out before training"
        from bigc_hip.train_step import mark
        mark("step glue (zero_grad ...)")
        packed = PackedBatch(list(proposal_list), self.pred_query_init.device, "lazy")      # the step picks the form of the feature rows
        mark("pack the batch")
        terms = HipTrainStep.apply(self, packed, list(gt_graph_list), *[p for _, p in self.named_parameters()])
        loss_dict = {"cls_pos": terms[0], "cls_neg": terms[1], "adj": terms[2]}
        return terms.sum(), loss_dict

    def invalidate_weights(self):
        """Call after modifying parameters through ``.data`` (in-place writes that do not bump the tensors' version counter,
        e.g. ``p.data.mul_()`` in hand-written EMA or weight-clipping code): the HIP library keeps split / re-laid-out copies of
        the weights and otherwise only refreshes them when a parameter moved or its version changed.  Also call it after
        REPLACING a parameter, buffer or submodule of a nested module by a new object (``model.decoder_layers[0].fc2 = ...``):
        the list of parameter tensors is walked once and kept (replacements on the model itself, ``.to()``, ``load_state_dict``
        and optimizer steps are noticed without it)."""
        object.__setattr__(self, "_state_items", None)       # also re-walk the module tree (see engine())
        if self._engine is not None:
            self._engine.invalidate_weights()

    def engine(self, force=False):
        """``force``: refresh the library's derived weight buffers even if no parameter's version counter moved (training steps:
        a fused optimizer updates the parameters without bumping it)"""
        if self._engine is None:
            self._engine = BigcEngine(self.config, self.variant)
        if self._state_items is None or not self._tree_unchanged():
            # (the state_dict walk over ~300 tensors costs 0.3 ms: kept while the module tree is what it was)
            self._state_items = list(self.state_dict(keep_vars=True).items())
            # what the cheap check below looks at: every module's place in its parent, every tensor's place in its module.  A nested
            # replacement -- ``load_state_dict(assign=True)``, ``model.encoder_layers[0].float()``, ``layer.fc = nn.Linear(..)``, a
            # parametrization -- changes one of these identities without passing through this class's __setattr__ / _apply.
            mods, tens = [], []
            for name, mod in self.named_modules():
                for cname, child in mod._modules.items():
                    mods.append((mod._modules, cname, child))
                for tname, t in mod._parameters.items():
                    tens.append((mod._parameters, tname, t))
                for tname, t in mod._buffers.items():
                    tens.append((mod._buffers, tname, t))
            object.__setattr__(self, "_tree_sig", (mods, tens))
        self._engine.sync_params(self._state_items, force)
        return self._engine

    def _tree_unchanged(self):
        """every module still sits where it sat in its parent and every parameter / buffer is still the same object (~40 us)"""
        mods, tens = self._tree_sig
        for d, k, v in mods:
            if d.get(k) is not v:
                return False
        for d, k, v in tens:
            if d.get(k) is not v:
                return False
        return True

    def __setattr__(self, name, value):
        if name != "_state_items" and isinstance(value, (torch.Tensor, nn.Module)):
            object.__setattr__(self, "_state_items", None)          # a parameter / buffer / submodule was replaced
        super().__setattr__(name, value)

    def _apply(self, fn, *args, **kwargs):
        object.__setattr__(self, "_state_items", None)              # .to() / .cuda() / .float() may swap the tensors
        return super()._apply(fn, *args, **kwargs)

    def register_buffer(self, *args, **kwargs):
        object.__setattr__(self, "_state_items", None)
        return super().register_buffer(*args, **kwargs)

    def register_parameter(self, *args, **kwargs):
        object.__setattr__(self, "_state_items", None)
        return super().register_parameter(*args, **kwargs)

    def forward_packed(self, packed, topk=None, keep_intermediates=False):
        """HIP forward on a resident PackedBatch -> dict of padded device tensors (no host sync)."""
        if topk is None:
            topk = self.default_topk
        with torch.no_grad():
            return self.engine().forward(packed, topk, True, keep_intermediates)

    @staticmethod
    def counts(out):
        """The one D2H copy (and host sync) of an eval step: the row offsets of the compacted results (-> triplet row counts per
        video).  The same copy carries the range flag of the split-operand GEMM path (csrc/gemm_ps.hip): an operand that its f16
        pair format cannot hold (|x| >= 4096, NaN, Inf) invalidates the step -> ``OverflowError``; ``forward`` then repeats the
        step on the exact f32 kernels."""
        if "_host_counts" not in out:
            both = out.pop("_host_pending")() if "_host_pending" in out else out["_counts"].cpu()
            out["_host_off"] = both[:-1]
            out["_host_counts"] = both[1:-1] - both[:-2]
            if int(both[-1]) != 0:
                raise OverflowError("an operand left the range of the split-operand GEMM: repeat the step with bigc_linear_set_exact(1)")
        return out["_host_counts"]

    def forward_checked(self, packed, topk):
        """forward_packed + the count copy; a step whose range flag is raised is repeated on the exact f32 GEMM kernels"""
        out = self.forward_packed(packed, topk)
        try:
            self.counts(out)
        except OverflowError:
            eng = self.engine()
            prev = eng.mode
            eng.set_mode(1, prev[1])                   # THIS handle only: other models / threads of the process keep their mode
            try:
                out = self.forward_packed(packed, topk)
                self.counts(out)
            finally:
                eng.set_mode(*prev)
        return out

    @staticmethod
    def unpack_results(n_total, keep, out, lazy=False):
        """Device outputs -> the reference's per-video list: one D2H copy (the row offsets) and one split per array -- the library has
        already laid the videos' rows out back to back in the reference's dtypes (``bigc_compact_triplets``).  ``lazy`` (keep = None: every
        video of the batch, in order): a list that cuts a video's four views out of the arrays when the video's entry is first read."""
        sizes = BigcBase.counts(out).tolist()
        total = int(out["_host_off"][-1])
        arrays = [out[k][:total] for k in ("c_quint", "c_scores", "c_dura", "c_qid")]
        if lazy:
            assert keep is None and len(sizes) == n_total
            return _LazyResults(arrays, out["_host_off"].tolist(), sizes)
        parts = [a.split(sizes) for a in arrays]
        res = [None] * n_total
        for j, i in enumerate(keep):
            if sizes[j] > 0:
                res[i] = (parts[0][j], parts[1][j], parts[2][j], parts[3][j])
        return res
