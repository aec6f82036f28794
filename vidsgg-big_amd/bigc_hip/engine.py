"""Host-side driver of the HIP BIG-C forward: owns the C model handle, keeps the parameter
table in sync with the torch parameters, allocates outputs/workspace with torch and calls
``bigc_forward`` on the current HIP stream."""
import ctypes
import os

import torch

from . import _lib
from .packed import PackedBatch


class BigcEngine(object):
    def __init__(self, cfg, variant):
        """``cfg``: the reference's ``model_config`` dict; ``variant``: "vidvrd" | "vidor"."""
        self.lib = _lib.load()
        c = _lib.bigc_config()
        c.variant = {"vidvrd": 0, "vidor": 1, "pairwise": 2}[variant]
        c.num_enti_cats, c.num_pred_cats = cfg["num_enti_cats"], cfg["num_pred_cats"]
        c.dim_feat, c.dim_clsme = cfg["dim_feat"], cfg["dim_clsme"]
        if variant == "pairwise":       # Base_C has no transformer: only the front-end and the pair head (models/model_pairwise_baseline.py:43-71)
            c.dim_enti = c.dim_pred = c.dim_att = cfg["dim_enti"]
            c.dim_ffn, c.enco_pool_len, c.n_enco_layers, c.n_deco_layers = cfg["dim_ffn"], cfg["enco_pool_len"], 0, 0
            c.n_att_head, c.num_querys = cfg["dim_enti"] // 16, 1
        else:
            c.dim_enti, c.dim_pred, c.dim_att, c.dim_ffn = cfg["dim_enti"], cfg["dim_pred"], cfg["dim_att"], cfg["dim_ffn"]
            c.enco_pool_len, c.n_enco_layers, c.n_deco_layers = cfg["enco_pool_len"], cfg["n_enco_layers"], cfg["n_deco_layers"]
            c.n_att_head, c.num_querys = cfg["n_att_head"], cfg["num_querys"]
        if variant == "vidvrd":
            c.dim_extra = cfg.get("dim_i3d", None) or 0
            c.use_clsme, c.has_enti_emb = 1, 1
        else:
            c.use_clsme = 1 if cfg["use_clsme"] else 0
            c.has_enti_emb = 1 if cfg.get("EntiNameEmb_path", None) is not None else 0
            c.dim_extra = cfg["dim_clsme"] if (c.use_clsme and not c.has_enti_emb) else 0
        self.c = c
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.bigc_create(ctypes.byref(c), ctypes.byref(self.handle)))
        self._sig = None
        self._dirty = False
        self._keep = None
        self._ws = None
        self._train_ws = None
        self.last_flat_grad = None
        self.mode = (-1, -1)

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle:
                self.lib.bigc_destroy(self.handle)
        except Exception:
            pass

    def sync_params(self, named_tensors, force=False):
        """(Re-)register the parameter pointers when a tensor moved, and rebuild the derived device buffers (``bigc_finalize``: new
        weights epoch) when a tensor was written in place.  ``force``: rebuild even though no version counter moved -- the training
        steps pass it, because the optimizer they follow may be a FUSED one (``torch.optim.Adam(fused=True)``), whose in-place
        update does not bump ``_version`` (measured, torch 2.10): without it every GEMM of the next forward would multiply by the
        split copy of the previous weights."""
        items = [(k, t) for k, t in named_tensors]
        ptrs = tuple((k, t.data_ptr()) for k, t in items)
        sig = (ptrs, tuple(t._version for _, t in items))
        if sig == self._sig and not force and not self._dirty:
            return
        if self._sig is None or self._sig[0] != ptrs:       # a training step changes the versions only: nothing to re-register
            for k, t in items:
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), \
                    "parameter %s must be a contiguous fp32 tensor on the GPU" % k
                _lib.check(self.lib.bigc_set_param(self.handle, k.encode(), t.data_ptr(), t.numel()))
        _lib.check(self.lib.bigc_finalize(self.handle, _lib.current_stream()))
        self._sig = sig
        self._dirty = False
        self._keep = [t for _, t in items]

    def invalidate_weights(self):
        """Force the next forward to re-register the parameters and rebuild every derived device buffer (split / transposed
        weight copies, ``role_cat``, the tap-major conv weight).  ``sync_params`` notices a parameter that moved or whose
        ``_version`` changed (optimizer steps, ``load_state_dict``, ``copy_``); it cannot notice writes through ``p.data``
        (``p.data.mul_()``, hand-written EMA / clipping code) or by a fused optimizer (``Adam(fused=True)`` updates the parameters
        without bumping ``_version``), which leave the version counter alone: call this after those.  (The training steps of this
        package refresh the weights on every forward themselves.)"""
        self._sig = None

    def weights_will_change(self):
        """Called at the end of every training backward of this package: an optimizer step follows, and a FUSED one
        (``torch.optim.Adam(fused=True)``) updates the parameters without bumping their version counters.  The next ``engine()`` call
        of ANY kind -- a training forward, but also an in-process validation forward, ``forward_packed``, the pairwise model -- then
        rebuilds the derived weight buffers (split / transposed copies, ``role_cat``, ``qkpos``) instead of multiplying by the
        copies of the previous weights.  The parameter pointers stay registered (only ``bigc_finalize`` runs again)."""
        self._dirty = True

    # ---- the error flag of the device matching (bigc_hungarian): scipy raises ValueError on a cost matrix with NaN / -inf or without
    # an assignment; a step matched on the device does not wait for the GPU, so the flag travels behind the step into a pinned word
    # (``bigc_stamp``: word 0 = number of the last step the GPU has finished matching, word 1 = a step at which the flag was up) and is
    # looked at when the next step starts (or when somebody reads ``model.last_match_indices``)
    def match_flag(self, device):
        if getattr(self, "_match_flag", None) is None or self._match_flag.device != torch.device(device):
            self._match_flag = torch.zeros((1,), dtype=torch.int32, device=device)
            self._match_words = torch.zeros((2,), dtype=torch.int64).pin_memory()
            self._match_np = self._match_words.numpy()
            self._match_seq = 0
        return self._match_flag

    def queue_match_flag_check(self, device):
        self._match_seq += 1
        with torch.cuda.device(device):
            _lib.check(self.lib.bigc_stamp(_lib.current_stream(), self._match_words.data_ptr(), self._match_seq, self._match_flag.data_ptr()))

    def check_match_flags(self, wait=False):
        if getattr(self, "_match_flag", None) is None:
            return
        if wait and self._match_np[0] < self._match_seq:
            torch.cuda.synchronize(self._match_flag.device)       # the marks are kernels on the step's stream: after this they have run
            if self._match_np[0] < self._match_seq:
                raise _lib.BigcError("the progress mark of a device-matched training step did not arrive (bigc_stamp)")
        if self._match_np[1] != 0:
            self._match_flag.zero_()
            self._match_np[1] = 0
            raise ValueError("matrix contains invalid numeric entries, or cost matrix is infeasible (linear_sum_assignment of a training "
                             "step matched on the device; BIGC_MATCH_ON_HOST=1 raises inside the step itself)")

    def flat_gradients(self, trainable, device, set_grad="bigc_set_grad", rebase="bigc_rebase_grads"):
        """A fresh zeroed flat fp32 buffer holding one gradient per entry of ``trainable`` [(key, parameter)] back to back, and
        the per-parameter views of it, registered as the gradient destinations of the C model.  The buffer is new every step
        (autograd may keep the views as ``.grad``); the layout is registered once and moved with one call afterwards."""
        layout = tuple((k, p.numel()) for k, p in trainable)
        total = sum(n for _, n in layout)
        flat = torch.zeros((total,), dtype=torch.float32, device=device)
        views = torch._utils._unflatten_dense_tensors(flat, [p for _, p in trainable])
        prev = getattr(self, "_grad_layout", None)
        if prev is not None and prev[0] == layout and rebase is not None:
            _lib.check(getattr(self.lib, rebase)(self.handle, prev[1], flat.data_ptr(), total))
        else:
            for k in set(k for k, _ in (prev[0] if prev else ())) - set(k for k, _ in layout):
                _lib.check(getattr(self.lib, set_grad)(self.handle, k.encode(), None))       # frozen since the last step
            off = 0
            for (k, n), v in zip(layout, views):
                assert v.data_ptr() == flat.data_ptr() + 4 * off
                _lib.check(getattr(self.lib, set_grad)(self.handle, k.encode(), v.data_ptr()))
                off += n
        self._grad_layout = (layout, flat.data_ptr())
        self.last_flat_grad = flat
        return flat, dict(zip((k for k, _ in trainable), views))

    def set_mode(self, exact=-1, fast=-1):
        """This handle's arithmetic mode (``bigc_set_mode``): ``exact`` = 1 every GEMM / attention kernel on the f32 MFMA, ``fast`` = 1
        the opt-in reduced-precision GEMMs, -1 = follow the process-wide default (BIGC_LINEAR_EXACT / BIGC_FAST,
        ``bigc_linear_set_exact`` / ``bigc_linear_set_fast``).  Other handles are not affected."""
        _lib.check(self.lib.bigc_set_mode(self.handle, int(exact), int(fast)))
        self.mode = (int(exact), int(fast))

    def set_dropout(self, p, seed):
        """Training-mode dropout of the transformer layers (0 = off); masks are a pure function of (seed, site, index)."""
        _lib.check(self.lib.bigc_set_dropout(self.handle, float(p), int(seed)))

    def forward(self, packed, topk, with_triplets=True, keep_intermediates=False):
        """Run a1..a8 on a PackedBatch. Returns a dict of device tensors (padded triplet arrays +
        ``n_out`` counts); nothing here synchronises with the host."""
        assert isinstance(packed, PackedBatch)
        c = self.c
        assert packed.feat_width == c.dim_feat + c.dim_extra, \
            "feature width %d != dim_feat+extra %d" % (packed.feat_width, c.dim_feat + c.dim_extra)
        dev = packed.device
        B, NT, Q, P = packed.n_vid, packed.n_trk, c.num_querys, c.num_pred_cats
        cap = Q * topk
        out = {
            "pred_logits": torch.empty((B * Q, P), dtype=torch.float32, device=dev),
            "att": torch.empty((2 * Q * NT,), dtype=torch.float32, device=dev),
            "soid": torch.empty((B, 2, Q), dtype=torch.int32, device=dev),
        }
        if keep_intermediates:
            out["pred_query"] = torch.empty((B * Q, c.dim_pred), dtype=torch.float32, device=dev)
            out["enti2enco"] = torch.empty((NT, c.dim_enti), dtype=torch.float32, device=dev)
            out["enco_output"] = torch.empty((NT, c.dim_enti), dtype=torch.float32, device=dev)
        if with_triplets:
            out["quint"] = torch.empty((B, cap, 5), dtype=torch.int32, device=dev)
            out["scores"] = torch.empty((B, cap, 3), dtype=torch.float32, device=dev)
            out["dura"] = torch.empty((B, cap, 2), dtype=torch.int32, device=dev)
            out["qid"] = torch.empty((B, cap), dtype=torch.int32, device=dev)
            out["n_out"] = torch.zeros((B,), dtype=torch.int32, device=dev)
            # row offsets of the compacted results and, behind them, the range flag of the split-operand GEMM path: ONE D2H copy returns both
            out["_counts"] = torch.zeros((B + 2,), dtype=torch.int32, device=dev)
            out["range_flag"] = out["_counts"][B + 1:]
            # the reference's return types, all videos back to back (bigc_compact_triplets): rows [off[v], off[v+1]) belong to video v
            out["c_quint"] = torch.empty((B * cap, 5), dtype=torch.int64, device=dev)
            out["c_scores"] = torch.empty((B * cap, 3), dtype=torch.float32, device=dev)
            out["c_dura"] = torch.empty((B * cap, 2), dtype=torch.int64, device=dev)
            out["c_qid"] = torch.empty((B * cap,), dtype=torch.int64, device=dev)
        o = _lib.bigc_outputs()
        for name, _ in _lib.bigc_outputs._fields_:
            setattr(o, name, out[name].data_ptr() if name in out else None)
        # The default eval forward reads the visual features in the split form: a batch that has no feature rows yet is gathered
        # straight into it (one launch, csrc/ingest.hip); every other mode reads f32 rows.  BIGC_FEAT_HR8=0 switches the form off.
        # A batch that holds no feature rows yet (forward(proposal_list): seen once) is read IN PLACE -- a table of row addresses, no copy;
        # a batch that already holds f32 rows (a resident PackedBatch, reused step after step) gets the split form once.
        use_hr8 = bool(with_triplets and os.environ.get("BIGC_FEAT_HR8", "1") != "0" and self.lib.bigc_forward_reads_hr8(self.handle))
        if use_hr8:
            if packed._feats is None and packed.feats_hr8 is None and packed.can_read_in_place() and c.dim_feat % 4 == 0 and c.dim_extra % 4 == 0 \
                    and os.environ.get("BIGC_FEAT_IN_PLACE", "1") != "0":
                packed.rows_in_place()
            else:
                packed.materialize("hr8", c.dim_feat)
        b = packed.as_struct(need_f32=not use_hr8)
        need = self.lib.bigc_workspace_bytes(self.handle, ctypes.byref(b))
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        _lib.check(self.lib.bigc_forward(self.handle, _lib.current_stream(), ctypes.byref(b), self._ws.data_ptr(),
                                         self._ws.numel(), int(topk), ctypes.byref(o)))
        if with_triplets:
            _lib.check(self.lib.bigc_compact_triplets(_lib.current_stream(), out["quint"].data_ptr(), out["scores"].data_ptr(), out["dura"].data_ptr(),
                                                      out["qid"].data_ptr(), out["n_out"].data_ptr(), B, cap, out["c_quint"].data_ptr(),
                                                      out["c_scores"].data_ptr(), out["c_dura"].data_ptr(), out["c_qid"].data_ptr(),
                                                      out["_counts"].data_ptr()))
        return out

    def pair_forward(self, packed, keep_intermediates=False):
        """Base_C: logits of every ordered tracklet pair of every video of a PackedBatch.
        -> dict(pair_logits (n_pairs, P), vid_pair_off host list, optional enti2enco)."""
        import numpy as np
        c = self.c
        assert packed.feat_width == c.dim_feat + c.dim_extra, "feature width %d != dim_feat+extra %d" % (packed.feat_width, c.dim_feat + c.dim_extra)
        dev = packed.device
        n = np.diff(packed.vid_trk_off_host).astype(np.int64)
        off = np.concatenate([[0], np.cumsum(n * (n - 1))])
        n_pairs = int(off[-1])
        out = {"vid_pair_off": off, "pair_logits": torch.empty((n_pairs, c.num_pred_cats), dtype=torch.float32, device=dev)}
        if n_pairs == 0:
            return out
        if keep_intermediates:
            out["enti2enco"] = torch.empty((packed.n_trk, c.dim_enti), dtype=torch.float32, device=dev)
        off_d = torch.from_numpy(off.astype(np.int32)).to(dev)
        b = packed.as_struct()
        need = self.lib.bigc_pair_workspace_bytes(self.handle, ctypes.byref(b), n_pairs)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        _lib.check(self.lib.bigc_pair_forward(self.handle, _lib.current_stream(), ctypes.byref(b), off_d.data_ptr(), n_pairs, self._ws.data_ptr(),
                                              self._ws.numel(), _lib.ptr(out.get("enti2enco")), out["pair_logits"].data_ptr()))
        out["_keepalive"] = off_d
        return out

    def pair_triplets(self, packed, pair_out, topk, rt_topk):
        """Base_C.construct_triplet for every video of the batch in the HIP library (csrc/pair_triplet.hip).
        -> list (one per video) of None or (quintuples i64 (U,5), scores f32 (U,3), dura_inters i64 (U,2)); ONE D2H copy (the row
        counts) for the whole batch (a second one, the durations, only when some video has no row at all: None vs empty)."""
        import numpy as np
        c = self.c
        dev = packed.device
        off = pair_out["vid_pair_off"]
        n_pairs = int(off[-1])
        B = packed.n_vid
        if n_pairs == 0:
            return [None] * B
        n = np.diff(packed.vid_trk_off_host).astype(np.int64)
        words = (n * n + 31) // 32
        bm_off = np.concatenate([[0], np.cumsum(words * c.num_pred_cats)])
        total_words = int(bm_off[-1])
        P = c.num_pred_cats
        off_d = pair_out["_keepalive"]
        bm_off_d = torch.from_numpy(bm_off[:-1].astype(np.int64)).to(dev)
        bm_words_d = torch.from_numpy(words.astype(np.int32)).to(dev)
        need = self.lib.bigc_pair_triplets_workspace_bytes(B, packed.n_trk, n_pairs, int(topk), total_words)
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        cap = n_pairs * topk
        quint = torch.empty((cap, 5), dtype=torch.int32, device=dev)
        scores = torch.empty((cap, 3), dtype=torch.float32, device=dev)
        dura = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        R = int(rt_topk) if rt_topk and rt_topk > 0 else 0
        counts = torch.zeros((2, B), dtype=torch.int32, device=dev)
        qt = torch.empty((B, max(R, 1), 5), dtype=torch.int32, device=dev)
        st = torch.empty((B, max(R, 1), 3), dtype=torch.float32, device=dev)
        dt = torch.empty((B, max(R, 1), 2), dtype=torch.int32, device=dev)
        _lib.check(self.lib.bigc_pair_triplets(_lib.current_stream(), pair_out["pair_logits"].data_ptr(), P, int(topk), packed.vid_trk_off.data_ptr(),
                                               off_d.data_ptr(), packed.trk_dura.data_ptr(), packed.trk_cat.data_ptr(), packed.trk_score.data_ptr(),
                                               B, packed.n_trk, n_pairs, bm_off_d.data_ptr(), bm_words_d.data_ptr(), total_words, ws.data_ptr(),
                                               ws.numel(), quint.data_ptr(), scores.data_ptr(), dura.data_ptr(), counts[0].data_ptr(), R,
                                               qt.data_ptr(), st.data_ptr(), dt.data_ptr(), counts[1].data_ptr()))
        cnt = counts.cpu().numpy()
        # The reference returns None only when NO ordered pair of the video overlaps in time (models/model_pairwise_baseline.py:373-377);
        # a video whose every surviving row is the background predicate gets a tuple of EMPTY tensors (:379-395).
        dura_h = None                                     # (downloaded only when some video produced no row: a second D2H copy otherwise saved)
        voff = packed.vid_trk_off_host
        res = []
        for v in range(B):
            if cnt[0, v] == 0:
                if dura_h is None:
                    dura_h = packed.trk_dura.cpu().numpy()
                d = dura_h[voff[v]:voff[v + 1]]
                ov = (np.maximum(d[:, None, 0], d[None, :, 0]) <= np.minimum(d[:, None, 1], d[None, :, 1]))
                np.fill_diagonal(ov, False)
                if not ov.any():
                    res.append(None)
                else:
                    res.append((torch.empty((0, 5), dtype=torch.long, device=dev), torch.empty((0, 3), dtype=torch.float32, device=dev),
                                torch.empty((0, 2), dtype=torch.long, device=dev)))
            elif R:
                m = int(cnt[1, v])
                res.append((qt[v, :m].long(), st[v, :m], dt[v, :m].long()))
            else:
                r0, m = int(off[v]) * topk, int(cnt[0, v])
                res.append((quint[r0:r0 + m].long(), scores[r0:r0 + m], dura[r0:r0 + m].long()))
        return res
