"""Host-side driver of the HIP grounding scorer (``grd_*`` entry points of include/bigc_hip.h)."""
import ctypes

import numpy as np
import torch

from . import _lib


def pos_table_rows(length, d):
    """(length, d) PosEncoder table, evaluated on the host in float32 exactly as the reference does
    (models/grd_model_v5.py:58-78: sin(pos * freq + phase) with python-double freq/phase cast to f32)."""
    i = np.arange(d)
    even = (i % 2) == 0
    freq = np.where(even, 10000.0 ** (-i / d), -(10000.0 ** ((1 - i) / d)))
    phase = np.where(even, 0.0, np.pi / 2)
    f = torch.from_numpy(freq.astype(np.float32))[None, :]
    ph = torch.from_numpy(phase.astype(np.float32))[None, :]
    pos = torch.arange(length, dtype=torch.float32)[:, None]
    return torch.sin(pos * f + ph).contiguous()


class GrdEngine(object):
    def __init__(self, cfg, n_enti_emb=81, n_pred_emb=51):
        self.lib = _lib.load()
        c = _lib.grd_config()
        c.dim_feat, c.dim_clsme, c.dim_hidden, c.num_bins = cfg["dim_feat"], cfg["dim_clsme"], cfg["dim_hidden"], cfg["num_bins"]
        c.n_enti_emb, c.n_pred_emb = n_enti_emb, n_pred_emb
        self.c = c
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.grd_create(ctypes.byref(c), ctypes.byref(self.handle)))
        self._sig = None
        self._keep = None
        self._ws = None
        self._tables = {}
        self._clip_cat = {}

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle:
                self.lib.grd_destroy(self.handle)
        except Exception:
            pass

    def sync_params(self, named_tensors, force=False):
        """``force``: see BigcEngine.sync_params (a fused optimizer's update does not bump the version counters)"""
        items = [(k, t) for k, t in named_tensors]
        sig = tuple((k, t.data_ptr(), t._version) for k, t in items)
        if sig == self._sig and not force and not getattr(self, "_dirty", False):
            return
        for k, t in items:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), \
                "parameter %s must be a contiguous fp32 tensor on the GPU" % k
            _lib.check(self.lib.grd_set_param(self.handle, k.encode(), t.data_ptr(), t.numel()))
        _lib.check(self.lib.grd_finalize(self.handle, _lib.current_stream()))
        self._sig = sig
        self._dirty = False
        self._keep = [t for _, t in items]

    def weights_will_change(self):
        """see BigcEngine.weights_will_change: called at the end of the training backward, so that the next forward of any kind
        re-finalizes even when a fused optimizer left the version counters alone"""
        self._dirty = True

    def set_mode(self, exact=-1, fast=-1):
        """this handle's arithmetic mode (``grd_set_mode``), see BigcEngine.set_mode"""
        _lib.check(self.lib.grd_set_mode(self.handle, int(exact), int(fast)))

    def invalidate_weights(self):
        """see BigcEngine.invalidate_weights: needed after writes through ``p.data``, which ``sync_params`` cannot see"""
        self._sig = None

    def tables(self, T, device):
        key = (T, str(device))
        if key not in self._tables:
            d = self.c.dim_hidden
            self._tables[key] = (pos_table_rows(T, d).to(device), pos_table_rows(3, d).to(device),
                                 torch.linspace(0, 1, T).to(device))      # clip_range (:705), host-evaluated
        return self._tables[key]

    def forward(self, video_feature, quint, dura, video_len, score_th, tiou_th, bins_th, nms_th, keep_intermediates=False):
        """video_feature (T,dim_feat) f32 cuda; quint (n,5), dura (n,2) integer tensors. -> dict of device tensors."""
        dev = video_feature.device
        assert video_feature.is_cuda and video_feature.dtype == torch.float32
        video_feature = video_feature.contiguous()
        T, n = int(video_feature.shape[0]), int(quint.shape[0])
        assert video_feature.shape[1] == self.c.dim_feat, "clip feature width %d != dim_feat %d" % (video_feature.shape[1], self.c.dim_feat)
        assert quint.shape == (n, 5) and dura.shape == (n, 2)
        q32 = quint.to(device=dev, dtype=torch.int32).contiguous()
        d32 = dura.to(device=dev, dtype=torch.int32).contiguous()
        k, d = self.c.num_bins, self.c.dim_hidden
        out = {
            "pooled_se": torch.empty((n, k + 1, 2), dtype=torch.float32, device=dev),
            "bins_probs": torch.empty((n, k + 1), dtype=torch.float32, device=dev),
            "bins_mask": torch.empty((n, k + 1), dtype=torch.uint8, device=dev),
            "status": torch.zeros((1,), dtype=torch.int32, device=dev),
        }
        if keep_intermediates:
            out["regrs"] = torch.empty((n * T, 2 * k), dtype=torch.float32, device=dev)
            out["conf"] = torch.empty((n * T, k), dtype=torch.float32, device=dev)
            out["cls"] = torch.empty((n * T, k), dtype=torch.float32, device=dev)
            out["video_enc"] = torch.empty((T, d), dtype=torch.float32, device=dev)
            out["query_enc"] = torch.empty((3 * n, d), dtype=torch.float32, device=dev)
            out["combined_enc"] = torch.empty((n * T, d), dtype=torch.float32, device=dev)
        o = _lib.grd_outputs()
        for name, _ in _lib.grd_outputs._fields_:
            setattr(o, name, out[name].data_ptr() if name in out else None)
        pos_v, pos_q, clip = self.tables(T, dev)
        need = self.lib.grd_workspace_bytes(self.handle, T, n)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        _lib.check(self.lib.grd_forward(self.handle, _lib.current_stream(), video_feature.data_ptr(), T, q32.data_ptr(), d32.data_ptr(), n,
                                        float(video_len), pos_v.data_ptr(), pos_q.data_ptr(), clip.data_ptr(),
                                        float(score_th), float(tiou_th), float(bins_th), float(nms_th),
                                        self._ws.data_ptr(), self._ws.numel(), ctypes.byref(o)))
        out["_keepalive"] = (q32, d32, video_feature)
        return out

    # rows of the combined encoder per call of forward_batch: 8 buffers of (rows, d) + (rows, 4d) + (rows, 3d) f32 ~ 8 KB per row at d=128
    BATCH_ROWS = 400000

    def forward_batch(self, jobs, score_th, tiou_th, bins_th, nms_th):
        """Several videos per launch sequence (``grd_forward_batch``).  ``jobs``: list of (video_feature (T,dim_feat) f32 cuda,
        quint (n,5), dura (n,2), video_len).  -> list of per-video dicts (views of the batch outputs) with the keys of ``forward``;
        ``status`` is a 1-element view per video.  Videos are grouped into calls of at most ``BATCH_ROWS`` combined rows."""
        # At head dims 8 / 32 / 64 the attention kernels keep the scores of a sequence in registers up to a head-dim dependent length and
        # switch to a two-pass form beyond it (csrc/attention.hip: bigc_mha_core_drop_f32), chosen per LAUNCH from the longest sequence:
        # there a video is only batched with videos on its own side of that limit, so that its result does not depend on its company
        # (identical, bit for bit, to the one-video call).  At head dim 16 (config G: 8 heads of 16) every sequence up to 1024 clips
        # runs the same chunked arithmetic (mha_core_hd16_lds_kernel): one class.
        hd = self.c.dim_hidden // 8
        limit = 1024 if hd == 16 else 32 * {8: 8, 32: 6, 64: 6}[hd]
        order = sorted(range(len(jobs)), key=lambda i: int(jobs[i][0].shape[0]) > limit)          # stable: short videos first, long ones after
        groups, cur, rows, side = [], [], 0, None
        for i in order:
            j = jobs[i]
            r = int(j[0].shape[0]) * int(j[1].shape[0])
            s_ = int(j[0].shape[0]) > limit
            if cur and (rows + r > self.BATCH_ROWS or s_ != side):
                groups.append(cur)
                cur, rows = [], 0
            cur.append(i)
            rows += r
            side = s_
        if cur:
            groups.append(cur)
        out = [None] * len(jobs)
        for g in groups:
            for i, o in zip(g, self._forward_group([jobs[i] for i in g], score_th, tiou_th, bins_th, nms_th)):
                out[i] = o
        return out

    def _forward_group(self, jobs, score_th, tiou_th, bins_th, nms_th):
        dev = jobs[0][0].device
        V = len(jobs)
        T = np.asarray([int(j[0].shape[0]) for j in jobs], dtype=np.int64)
        nq = np.asarray([int(j[1].shape[0]) for j in jobs], dtype=np.int64)
        assert T.min() > 0 and nq.min() > 0, "videos without clips or without queries are filtered out by the caller"
        Q, total_T, total_rows, max_T = int(nq.sum()), int(T.sum()), int((T * nq).sum()), int(T.max())
        k, d = self.c.num_bins, self.c.dim_hidden
        feat = torch.cat([j[0].to(dtype=torch.float32) for j in jobs], 0).contiguous()
        assert feat.is_cuda and feat.shape[1] == self.c.dim_feat
        # (concatenate first, convert once: one conversion kernel per table instead of one per video)
        q32 = torch.cat([j[1].to(device=dev) for j in jobs], 0).to(torch.int32).contiguous()
        d32 = torch.cat([j[2].to(device=dev) for j in jobs], 0).to(torch.int32).contiguous()
        q_vid = np.repeat(np.arange(V), nq)
        tables = [np.concatenate([[0], np.cumsum(T)]), q_vid, np.concatenate([[0], np.cumsum(T[q_vid])])]
        sizes = [(len(a) + 3) // 4 * 4 for a in tables]
        starts = np.concatenate([[0], np.cumsum(sizes)])
        blob = np.zeros((int(starts[-1]),), dtype=np.int32)
        for a, o in zip(tables, starts):
            blob[o:o + len(a)] = a
        blob_d = _lib.upload(blob, dev)
        vid_row_off, q_vid_d, q_row_off = [blob_d[o:o + len(a)] for a, o in zip(tables, starts)]
        vlen = _lib.upload(np.asarray([float(j[3]) for j in jobs], dtype=np.float32), dev)
        pos_v, pos_q, _ = self.tables(max_T, dev)
        ckey = (tuple(int(t) for t in T), str(dev))          # the clips' normalised positions of this group of videos, back to back
        clip = self._clip_cat.get(ckey)
        if clip is None:
            if len(self._clip_cat) >= 64:
                self._clip_cat.clear()
            clip = self._clip_cat[ckey] = torch.cat([self.tables(int(t), dev)[2] for t in T])
        b = _lib.grd_batch()
        b.n_vid, b.n_query, b.max_T, b.total_T, b.total_rows = V, Q, max_T, total_T, total_rows
        b.vid_row_off, b.q_vid, b.q_row_off, b.vid_len = vid_row_off.data_ptr(), q_vid_d.data_ptr(), q_row_off.data_ptr(), vlen.data_ptr()
        res = {
            "pooled_se": torch.empty((Q, k + 1, 2), dtype=torch.float32, device=dev),
            "bins_probs": torch.empty((Q, k + 1), dtype=torch.float32, device=dev),
            "bins_mask": torch.empty((Q, k + 1), dtype=torch.uint8, device=dev),
            "status": torch.empty((1 + V,), dtype=torch.int32, device=dev),          # (zeroed by grd_forward_batch on its stream)
        }
        o = _lib.grd_outputs()
        for name, _ in _lib.grd_outputs._fields_:
            setattr(o, name, res[name].data_ptr() if name in res else None)
        need = self.lib.grd_batch_workspace_bytes(self.handle, ctypes.byref(b))
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = None
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        _lib.check(self.lib.grd_forward_batch(self.handle, _lib.current_stream(), feat.data_ptr(), ctypes.byref(b), q32.data_ptr(), d32.data_ptr(),
                                              pos_v.data_ptr(), pos_q.data_ptr(), clip.data_ptr(), float(score_th), float(tiou_th),
                                              float(bins_th), float(nms_th), self._ws.data_ptr(), self._ws.numel(), ctypes.byref(o)))
        keep = (feat, q32, d32, blob_d, vlen, clip)
        qo = np.concatenate([[0], np.cumsum(nq)])
        return [{"pooled_se": res["pooled_se"][qo[v]:qo[v + 1]], "bins_probs": res["bins_probs"][qo[v]:qo[v + 1]],
                 "bins_mask": res["bins_mask"][qo[v]:qo[v + 1]], "status": res["status"][1 + v:2 + v], "_batch_status": res["status"],
                 "_keepalive": keep} for v in range(V)]
