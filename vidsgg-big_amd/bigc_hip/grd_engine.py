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

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle:
                self.lib.grd_destroy(self.handle)
        except Exception:
            pass

    def sync_params(self, named_tensors):
        items = [(k, t) for k, t in named_tensors]
        sig = tuple((k, t.data_ptr(), t._version) for k, t in items)
        if sig == self._sig:
            return
        for k, t in items:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), \
                "parameter %s must be a contiguous fp32 tensor on the GPU" % k
            _lib.check(self.lib.grd_set_param(self.handle, k.encode(), t.data_ptr(), t.numel()))
        _lib.check(self.lib.grd_finalize(self.handle, _lib.current_stream()))
        self._sig = sig
        self._keep = [t for _, t in items]

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
