"""Device-resident ragged tracklet store: the packed form of a list of ``TrajProposal``.

The reference keeps one small tensor per tracklet and stretches/stacks them inside
``forward`` (models/model_0v10.py:391-430).  Here the tracklets of a batch of videos are
gathered into flat HBM buffers plus int32 offset tables, which is what ``bigc_forward``
consumes (``bigc_batch`` in ``include/bigc_hip.h``).

Packing is ONE kernel launch (``bigc_pack_tracklets``, csrc/ingest.hip) over a table of
per-tracklet source addresses -- not a ``torch.cat`` over thousands of tensors -- and writes
the rows directly in the form the forward reads:

* ``"hr8"`` (the eval forward's form): the visual (RoI) columns as f16 head / residual pairs
  (the operand format of csrc/gemm_ps.hip, same footprint as f32) + the remaining columns
  (I3D / classeme) as an f32 tensor of their own.  A feature element exists once in HBM.
* ``"f32"``: full f32 rows (training, exact / bf16 arithmetic modes).
* in place (``rows_in_place``): NO copy -- a per-row address table through which the default eval forward reads the tracklets' rows
  where they are (one-shot batches: ``forward(proposal_list)``); only the boxes are gathered.

The form is chosen at first use (``materialize``); a packed batch can be kept resident and
reused.  Per-video metadata (categories, scores, durations) is gathered by a second small launch.
"""
import weakref

import numpy as np
import torch

from . import _lib


def tracklet_table(p, device):
    """Per-video source table of a TrajProposal-like object: tracklet lengths, addresses and row pitches of the feature rows,
    addresses of the boxes -- what ``bigc_pack_tracklets`` gathers from.  Cached on the object (``_bigc_pack``): containers
    built by this package (``TrajProposal.from_blocks`` / ``.to()``, ``bigc_hip.ingest``) hold their tracklets as views of one
    block per video and get the table by arithmetic; for foreign containers (the reference's own ``TrajProposal``) the
    tensors are visited once.  The cache is checked against EVERY item of both lists: it holds a weak reference per source
    tensor (whose callback marks the table dead when the tensor goes) and the tensors' ids, and is a hit only while every one is alive and
    IS the list's current item -- after the reference's to(GPU) -> forward -> to(cpu)
    -> to(GPU) cycle the caching allocator may hand the old blocks back for the first / last tracklets and other blocks for the middle
    ones; matching end addresses prove nothing.
    The cached table holds ADDRESSES and weak references only, never tensors (a table that kept its sources alive pinned every evaluated
    video's GPU features for the whole run once the reference's ``TrajProposal.to(cpu)`` had replaced the list items); what keeps the
    memory behind the addresses alive while a forward reads it is the ``PackedBatch`` (``_sources``), for as long as IT lives.  A table
    made for another device is dropped."""
    c = getattr(p, "_bigc_pack", None)
    if c is not None and c.get("device") != device:
        try:
            del p._bigc_pack
        except Exception:
            pass
        c = None
    if c is not None and "blocks" in c:
        # a container of this package: its tracklets ARE views of two blocks (no per-tracklet tensor objects exist)
        if c["device"] == device and c["blocks"] == (p._feat_block.data_ptr(), p._box_block.data_ptr()) and c["n"] == int(p.num_proposals):
            return c
        c = None
    fl, bl = p.features_list, p.bboxes_list
    n = len(bl)
    assert n == len(fl) and n == int(p.num_proposals)
    if c is not None and c["n"] == n and c["device"] == device and not c["died"][0] and c["ids"] == (list(map(id, fl)), list(map(id, bl))):
        # (every source tensor of the table is still alive -- a weak-reference callback sets `died` when one goes -- so its id() cannot have
        # been handed to another object: equal ids ARE the same tensors.  5 us per video where 128 weak-reference calls took 27.)
        # the caller's lists as they are NOW keep the rows alive for this batch (64 references per video; nothing is stored on `p`);
        # `_cached`: where PackedBatch keeps what it derives from the table (tmax, the metadata check) from one call to the next
        return dict(c, sources=(list(fl), list(bl)), _cached=c)
    keep = []
    width = int(fl[0].shape[1])

    def usable(t, cols, contiguous):
        assert t.dim() == 2 and t.shape[1] == cols
        if t.dtype != torch.float32 or t.device != device or t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < cols) or \
                (contiguous and not t.is_contiguous()):
            t = t.to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
        return t

    fl2, bl2 = [usable(f, width, False) for f in fl], [usable(b, 4, True) for b in bl]
    lens = np.fromiter((b.shape[0] for b in bl2), dtype=np.int64, count=n)
    flens = np.fromiter((f.shape[0] for f in fl2), dtype=np.int64, count=n)
    assert (lens == flens).all() and lens.min() > 0, "every tracklet needs as many feature rows as boxes"
    fptr = np.fromiter((f.data_ptr() for f in fl2), dtype=np.uint64, count=n)
    bptr = np.fromiter((b.data_ptr() for b in bl2), dtype=np.uint64, count=n)
    fld = np.fromiter((f.stride(0) if f.shape[0] > 1 else width for f in fl2), dtype=np.int64, count=n)
    died = [False]
    c = dict(n=n, device=device, width=width, lens=lens, fptr=fptr, bptr=bptr, fld=None if (fld == width).all() else fld.astype(np.int32),
             aligned=bool(((fptr | bptr) % np.uint64(16) == 0).all() and (fld % 4 == 0).all()),
             ids=(list(map(id, fl)), list(map(id, bl))), died=died, refs=[weakref.ref(t, lambda _r, d=died: d.__setitem__(0, True)) for t in fl + bl], keep=[])
    cached = None
    if not keep:                  # (a table that points at converted COPIES is not kept: the caller may change its own tensors' values in place)
        try:
            p._bigc_pack = c      # addresses + weak references only
            cached = c
        except Exception:
            pass
    # the copy handed to the PackedBatch carries the references that keep the addressed memory alive (a caller that drops a tensor from
    # the middle of its list while the batch is in flight must not leave the forward reading freed memory)
    return dict(c, keep=keep, sources=(list(fl2), list(bl2)), _cached=cached if cached is not None else {})


class PackedBatch(object):
    _meta_srcs = None

    def __init__(self, proposals, device, features="f32"):
        """``proposals``: list of non-empty TrajProposal-like objects (see bigc_hip.synthetic.TrajProposal).
        ``features``: "f32" / "hr8" = gather the feature rows now in that form, "lazy" = at first use (the engine picks the form
        its forward reads; ``dim_feat`` is only known there)."""
        assert len(proposals) > 0
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.proposals = proposals
        n_per_vid = [int(p.num_proposals) for p in proposals]
        assert min(n_per_vid) > 0, "empty videos must be filtered out before packing"
        tabs = [tracklet_table(p, device) for p in proposals]
        self._keep = [t["keep"] for t in tabs if t["keep"]]
        self._sources = [t["sources"] for t in tabs if "sources" in t]      # foreign containers: their tensors, for the life of THIS batch
        self.feat_width = tabs[0]["width"]
        assert all(t["width"] == self.feat_width for t in tabs), "all videos of a batch need the same feature width"
        self.n_vid = len(proposals)
        self.n_trk = int(sum(n_per_vid))
        self.max_trk = int(max(n_per_vid))
        all_lens = np.concatenate([t["lens"] for t in tabs])
        for t in tabs:
            if "tmax" not in t:
                # kept with the CACHED table (256 small reductions per call otherwise): for a foreign container `t` is a per-call copy and
                # `_cached` the dict that lives on the proposal
                t["tmax"] = t.get("_cached", t)["tmax"] = int(t["lens"].max())
        tmax = np.asarray([t["tmax"] for t in tabs])
        wh = [[float(p.video_wh[0]), float(p.video_wh[1])] for p in proposals]
        self.n_rows = int(all_lens.sum())
        trk_vid = np.repeat(np.arange(self.n_vid), n_per_vid)
        t_o = (tmax - 1) // 2 + 1                                 # conv k3/s2/p1 output length
        self.n_conv = int((t_o[trk_vid]).sum())
        assert self.n_rows < 2 ** 31 and self.n_conv * 3 < 2 ** 31
        self.vid_trk_off_host = np.concatenate([[0], np.cumsum(n_per_vid)]).astype(np.int64)
        self._vectorized = all(t["aligned"] for t in tabs) and self.feat_width % 4 == 0

        # per-video metadata: gathered on the device when the containers hold what the reference's do (int64 / f32 / int64)
        def meta_usable(p, t):
            # the cache holds REFERENCES to the three (small) metadata tensors and compares identities: id() alone could come back for a new
            # object after the old one was freed; three data_ptr() calls per video cost 0.15 ms per 256-video call
            a, b_, c_ = p.cat_ids, p.scores, p.traj_durations
            key = t.get("meta_key")
            if key is None or key[0] is not a or key[1] is not b_ or key[2] is not c_:          # checked once per set of tensors
                store = t.get("_cached", t)
                store["meta_key"] = t["meta_key"] = (a, b_, c_)
                store["meta_ok"] = t["meta_ok"] = bool(p.cat_ids.dtype == torch.int64 and p.scores.dtype == torch.float32 and p.traj_durations.dtype == torch.int64
                                    and p.cat_ids.device == device and p.scores.device == device and p.traj_durations.device == device
                                    and p.cat_ids.is_contiguous() and p.scores.is_contiguous() and p.traj_durations.is_contiguous())
                store["meta_ptr"] = t["meta_ptr"] = (p.cat_ids.data_ptr(), p.scores.data_ptr(), p.traj_durations.data_ptr())
            return t["meta_ok"]
        meta_ok = device.type == "cuda" and all([meta_usable(p, t) for p, t in zip(proposals, tabs)])

        # every table in ONE host-to-device copy: [u64 feature addresses | u64 box addresses | u64 metadata addresses | int32 tables]
        any_ld = any(t["fld"] is not None for t in tabs)
        u64 = [np.concatenate([t["fptr"] for t in tabs]), np.concatenate([t["bptr"] for t in tabs])]
        if meta_ok:
            mp = np.asarray([t["meta_ptr"] for t in tabs], dtype=np.uint64)
            u64 += [mp[:, 0], mp[:, 1], mp[:, 2]]
        tables = [np.concatenate([[0], np.cumsum(all_lens)]), np.concatenate([[0], np.cumsum(t_o[trk_vid])]), trk_vid,
                  self.vid_trk_off_host, tmax]
        if any_ld:
            tables.append(np.concatenate([t["fld"] if t["fld"] is not None else np.full(t["n"], self.feat_width, np.int32) for t in tabs]))
        n64 = [(len(a) + 1) // 2 * 2 for a in u64]                # 16-byte aligned starts
        n32 = [(len(a) + 3) // 4 * 4 for a in tables]
        blob = np.zeros((8 * sum(n64) + 4 * sum(n32),), dtype=np.uint8)
        b64, b32 = blob[:8 * sum(n64)].view(np.uint64), blob[8 * sum(n64):].view(np.int32)
        s64, s32 = np.concatenate([[0], np.cumsum(n64)]), np.concatenate([[0], np.cumsum(n32)])
        for a, o in zip(u64, s64):
            b64[o:o + len(a)] = a
        for a, o in zip(tables, s32):
            b32[o:o + len(a)] = a
        blob_d = _lib.upload(blob, device)
        d64, d32 = blob_d[:8 * sum(n64)].view(torch.int64), blob_d[8 * sum(n64):].view(torch.int32)
        self._feat_src, self._box_src = d64[s64[0]:s64[0] + self.n_trk], d64[s64[1]:s64[1] + self.n_trk]
        views = [d32[o:o + len(a)] for a, o in zip(tables, s32)]
        self.trk_row_off, self.trk_conv_off, self.trk_vid, self.vid_trk_off, self.vid_tmax = views[:5]
        self._feat_ld = views[5] if any_ld else None
        self.vid_wh = _lib.upload(np.asarray(wh, dtype=np.float32).reshape(-1, 2), device)
        if meta_ok:
            self.trk_dura = torch.empty((self.n_trk, 2), dtype=torch.int32, device=device)
            self.trk_cat = torch.empty((self.n_trk,), dtype=torch.int32, device=device)
            self.trk_score = torch.empty((self.n_trk,), dtype=torch.float32, device=device)
            srcs = self._meta_srcs = [d64[s64[k]:s64[k] + self.n_vid] for k in (2, 3, 4)]
            _lib.check(_lib.load().bigc_pack_video_meta(_lib.current_stream(), srcs[0].data_ptr(), srcs[1].data_ptr(), srcs[2].data_ptr(),
                                                        self.vid_trk_off.data_ptr(), self.n_vid, self.trk_cat.data_ptr(), self.trk_score.data_ptr(),
                                                        self.trk_dura.data_ptr()))
        else:
            def on_device(t):
                return t if t.device == device else t.to(device)
            self.trk_dura = torch.cat([on_device(p.traj_durations) for p in proposals], 0).to(torch.int32).contiguous()
            self.trk_cat = torch.cat([on_device(p.cat_ids) for p in proposals], 0).to(torch.int32).contiguous()
            self.trk_score = torch.cat([on_device(p.scores) for p in proposals], 0).to(torch.float32).contiguous()
        self._blob = blob_d
        self._feats, self._boxes, self.extra = None, None, None
        self.feats_hr8, self.feats_hr8_flag = None, None
        self.feat_row_ptr = None
        if features != "lazy":
            assert features == "f32", "the split form needs dim_feat: PackedBatch(..., 'lazy') then materialize('hr8', dim_feat)"
            self.materialize("f32")

    # ---- the feature rows ---------------------------------------------------------------------------------------------------
    def materialize(self, form, dim_feat=None):
        """Gather the tracklets' rows (one launch; the boxes ride along the first time).  ``form`` "f32": full f32 rows;
        "hr8": the first ``dim_feat`` columns in the split form + the rest as ``extra`` (f32).  Idempotent; a batch that
        already holds f32 rows gets the split form from them (``bigc_split_rows_f32``), and one whose source containers were
        dropped (``proposals = None``) cannot be re-gathered."""
        dev, R, W = self.device, self.n_rows, self.feat_width
        if form == "f32":
            if self._feats is not None:
                return
        else:
            assert form == "hr8" and dim_feat is not None and dim_feat % 8 == 0 and dim_feat <= W
            if self.feats_hr8 is not None and self.feats_hr8.shape[1] == 4 * dim_feat:
                return
        if dev.type != "cuda":
            # host-only use (tests of the offset tables): no kernels off the GPU
            assert form == "f32", "the split form is written by the HIP library only"
            self._boxes = torch.cat([b for p in self.proposals for b in p.bboxes_list], 0).to(torch.float32).contiguous()
            self._feats = torch.cat([f for p in self.proposals for f in p.features_list], 0).to(torch.float32).contiguous()
            return
        lib = _lib.load()
        if form == "hr8" and self._feats is not None:
            self.feats_hr8 = torch.empty((R, 4 * dim_feat), dtype=torch.uint8, device=dev)
            self.feats_hr8_flag = torch.zeros((1,), dtype=torch.int32, device=dev)
            _lib.check(lib.bigc_split_rows_f32(_lib.current_stream(), self._feats.data_ptr(), W, self.feats_hr8.data_ptr(), 4 * dim_feat,
                                               R, dim_feat, self.feats_hr8_flag.data_ptr()))
            return
        assert self.proposals is not None, "the source containers of this batch were dropped: it cannot be gathered again in another form"
        boxes = None
        if self._boxes is None:
            boxes = torch.empty((R, 4), dtype=torch.float32, device=dev)
        feats = hr8 = extra = flag = None
        if form == "f32":
            feats = torch.empty((R, W), dtype=torch.float32, device=dev)
        else:
            hr8 = torch.empty((R, 4 * dim_feat), dtype=torch.uint8, device=dev)
            flag = torch.zeros((1,), dtype=torch.int32, device=dev)
            if W > dim_feat:
                extra = torch.empty((R, W - dim_feat), dtype=torch.float32, device=dev)
        vec = self._vectorized and (form == "f32" or (dim_feat % 4 == 0 and (W - dim_feat) % 4 == 0))
        _lib.check(lib.bigc_pack_tracklets(_lib.current_stream(), self._feat_src.data_ptr(), _lib.ptr(self._feat_ld), self._box_src.data_ptr(),
                                           self.trk_row_off.data_ptr(), self.n_trk, R, W, dim_feat if form == "hr8" else W, 1 if vec else 0,
                                           _lib.ptr(feats), W, _lib.ptr(hr8), 4 * (dim_feat or 0), _lib.ptr(extra), (W - dim_feat) if extra is not None else 0,
                                           _lib.ptr(boxes), _lib.ptr(flag)))
        if boxes is not None:
            self._boxes = boxes
        if form == "f32":
            self._feats = feats
        else:
            self.feats_hr8, self.feats_hr8_flag, self.extra = hr8, flag, extra

    # ---- the same list of containers again (models/_bigc.py: forward keeps its last batch) -----------------------------------------------
    def signature(self):
        """What must still hold for this batch's tables to describe ``self.proposals``: weak references to every container's two blocks
        and three metadata tensors (containers of this package only; None otherwise).  Weak: the signature keeps no GPU memory alive."""
        ps = self.proposals
        if ps is None or self._meta_srcs is None or self._sources or self._keep or not all(hasattr(p, "_feat_block") and p._lists is None for p in ps):
            return None
        return [(weakref.ref(p._feat_block), weakref.ref(p._box_block), weakref.ref(p.cat_ids), weakref.ref(p.scores), weakref.ref(p.traj_durations)) for p in ps]

    @staticmethod
    def signature_holds(sig, proposals):
        """every container still holds the very tensors the tables were made from (identity: ~0.3 us per container)"""
        for p, (fb, bb, ci, sc, td) in zip(proposals, sig):
            if fb() is not p._feat_block or bb() is not p._box_block or ci() is not p.cat_ids or sc() is not p.scores or td() is not p.traj_durations:
                return False
        return True

    def refresh_values(self):
        """The tables (addresses, offsets) are those of the same tensors; what the batch COPIED from them when it was made -- boxes and the
        per-tracklet metadata -- is gathered again (two launches), so that values written in place since then are seen, as the feature rows
        (read in place) are."""
        lib = _lib.load()
        if self._boxes is not None:
            _lib.check(lib.bigc_pack_tracklets(_lib.current_stream(), None, None, self._box_src.data_ptr(), self.trk_row_off.data_ptr(),
                                               self.n_trk, self.n_rows, self.feat_width, self.feat_width, 0, None, 0, None, 0, None, 0,
                                               self._boxes.data_ptr(), None))
        srcs = self._meta_srcs
        _lib.check(lib.bigc_pack_video_meta(_lib.current_stream(), srcs[0].data_ptr(), srcs[1].data_ptr(), srcs[2].data_ptr(),
                                            self.vid_trk_off.data_ptr(), self.n_vid, self.trk_cat.data_ptr(), self.trk_score.data_ptr(),
                                            self.trk_dura.data_ptr()))

    def can_read_in_place(self):
        """the forward can read the feature rows where they are (16-byte aligned sources that are still alive)"""
        return self.device.type == "cuda" and self._vectorized and self.proposals is not None

    def rows_in_place(self):
        """No copy of the features at all: the per-row address table (``bigc_batch.feat_row_ptr``) the default eval forward reads the
        callers' per-tracklet rows through.  The source tensors must stay alive (``proposals`` holds them)."""
        assert self.can_read_in_place()
        if self.feat_row_ptr is None:
            self.feat_row_ptr = torch.empty((self.n_rows,), dtype=torch.int64, device=self.device)
            _lib.check(_lib.load().bigc_row_pointers(_lib.current_stream(), self._feat_src.data_ptr(), _lib.ptr(self._feat_ld), self.feat_width,
                                                     self.trk_row_off.data_ptr(), self.n_trk, self.feat_row_ptr.data_ptr()))

    def split_features(self, dim_feat):
        """(kept name) the visual feature columns in the split form "hr8": see ``materialize``"""
        self.materialize("hr8", dim_feat)

    @property
    def feats(self):
        """(R, width) f32 rows, gathered at first use"""
        if self._feats is None:
            self.materialize("f32")
        return self._feats

    @property
    def boxes(self):
        if self._boxes is None:
            if self.device.type != "cuda":
                self.materialize("f32")
            else:
                self._boxes = torch.empty((self.n_rows, 4), dtype=torch.float32, device=self.device)
                _lib.check(_lib.load().bigc_pack_tracklets(_lib.current_stream(), None, None, self._box_src.data_ptr(), self.trk_row_off.data_ptr(),
                                                           self.n_trk, self.n_rows, self.feat_width, self.feat_width, 0, None, 0, None, 0, None, 0,
                                                           self._boxes.data_ptr(), None))
        return self._boxes

    def drop_sources(self):
        """forget the source containers (their tensors may then be freed); the gathered forms stay"""
        self.proposals, self._keep, self._sources = None, None, None

    def n_pairs(self):
        """tracklet-pairs of the batch: sum_v N_v (N_v - 1)."""
        n = np.diff(self.vid_trk_off_host)
        return int((n * (n - 1)).sum())

    def as_struct(self, need_f32=True):
        """``need_f32``: the call reads ``bigc_batch.feats`` (everything but the default eval forward, which reads the split form
        when the batch holds it, or the rows in place)"""
        if need_f32 or (self.feats_hr8 is None and self.feat_row_ptr is None):
            self.materialize("f32")
        b = _lib.bigc_batch()
        b.n_vid, b.n_trk, b.n_rows, b.n_conv, b.max_trk = self.n_vid, self.n_trk, self.n_rows, self.n_conv, self.max_trk
        for name in ("trk_row_off", "trk_conv_off", "trk_vid", "trk_dura", "trk_cat", "trk_score", "vid_trk_off", "vid_tmax", "vid_wh"):
            setattr(b, name, getattr(self, name).data_ptr())
        b.boxes = self.boxes.data_ptr()
        b.feats = self._feats.data_ptr() if self._feats is not None else None
        if self.feats_hr8 is not None:
            b.feats_hr8, b.ld_feats_hr8, b.feats_hr8_flag = self.feats_hr8.data_ptr(), int(self.feats_hr8.shape[1]), self.feats_hr8_flag.data_ptr()
            if self.extra is not None:
                b.extra, b.ld_extra = self.extra.data_ptr(), int(self.extra.shape[1])
        elif self.feat_row_ptr is not None and self._feats is None:
            b.feat_row_ptr = self.feat_row_ptr.data_ptr()
        return b
