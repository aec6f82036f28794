"""Temporal grounding scorer on the HIP path: drop-in for the reference's
``models/grd_model_v5.py:DEBUG`` (same constructor, ``state_dict`` keys and eval ``forward``).
The torch modules below only own parameters under the reference's key names; the arithmetic runs
in the HIP library (``bigc_hip.grd_engine.GrdEngine``)."""
import numpy as np
import torch
from torch import nn

from bigc_hip.grd_engine import GrdEngine
from bigc_hip.grd_train_step import LOSS_KEYS, HipGrdTrainStep, gt_queries


class _DWSep(nn.Module):
    """keys: depth_wise.{weight,bias}, point_wise.{weight,bias}  (models/grd_model_v5.py:41-45)"""

    def __init__(self, c_in, c_out, k):
        super().__init__()
        self.depth_wise = nn.Conv1d(c_in, c_in, k, 1, padding=k // 2, groups=c_in)
        self.point_wise = nn.Conv1d(c_in, c_out, kernel_size=1)
        nn.init.kaiming_normal_(self.depth_wise.weight)
        nn.init.kaiming_normal_(self.point_wise.weight)
        nn.init.constant_(self.depth_wise.bias, 0.0)
        nn.init.constant_(self.point_wise.bias, 0.0)


class _QANetParams(nn.Module):
    """keys: convs.{0-3}.*, mh_attn.*, fc, normb, norm_seq.{0-3}, norme  (models/grd_model_v5.py:86-97)"""

    def __init__(self, d, num_conv, k):
        super().__init__()
        self.convs = nn.ModuleList([_DWSep(d, d, k) for _ in range(num_conv)])
        self.mh_attn = nn.MultiheadAttention(d, num_heads=8, dropout=0.1)
        self.fc = nn.Linear(d, d, bias=True)
        self.normb = nn.LayerNorm(d)
        self.norm_seq = nn.ModuleList([nn.LayerNorm(d) for _ in range(num_conv)])
        self.norme = nn.LayerNorm(d)


def _head(d, c_out):
    layers = [nn.Sequential(_DWSep(d, d, 3), nn.ReLU()) for _ in range(4)]
    return layers + [_DWSep(d, c_out, 3)]


class DEBUG(nn.Module):
    def __init__(self, config, is_train=True):
        super().__init__()
        self.is_train = is_train
        self.config = dict(config)
        self.dim_feat, self.dim_clsme = config["dim_feat"], config["dim_clsme"]
        self.dim_hidden, self.num_bins = config["dim_hidden"], config["num_bins"]
        self.loss_factor = config["loss_factor"]
        enti = torch.from_numpy(np.load(config["EntiNameEmb_path"])).float()
        pred = torch.from_numpy(np.load(config["PredNameEmb_path"])).float()
        assert enti.shape == (81, 300) and pred.shape == (51, 300)       # models/grd_model_v5.py:162,168
        self.EntiNameEmb = nn.Parameter(enti, requires_grad=True)
        self.PredNameEmb = nn.Parameter(pred, requires_grad=True)
        d = self.dim_hidden
        self.video_fc = nn.Linear(self.dim_feat, d)
        self.query_fc = nn.Linear(self.dim_clsme, d)
        self.temp_fc = nn.Linear(2, d)
        self.vq_fc = nn.Linear(4 * d, d)
        self.video_encoder = _QANetParams(d, 4, 7)
        self.query_encoder = _QANetParams(d, 4, 3)
        self.combined_encoder = _QANetParams(d, 4, 7)
        self.proj2sim = nn.Linear(d, d, bias=False)
        self.cls_head = nn.Sequential(*_head(d, self.num_bins))
        self.conf_head = nn.Sequential(*_head(d, self.num_bins))
        self.regr_head = nn.Sequential(*(_head(d, 2 * self.num_bins) + [nn.Sigmoid()]))
        self._engine = None

    def invalidate_weights(self):
        """Call after modifying parameters through ``.data`` (in-place writes that do not bump the tensors' version counter,
        e.g. ``p.data.mul_()`` in hand-written EMA or weight-clipping code): the HIP library keeps split / re-laid-out copies of
        the weights and otherwise only refreshes them when a parameter moved or its version changed."""
        if self._engine is not None:
            self._engine.invalidate_weights()

    def engine(self, force=False):
        if self._engine is None:
            self._engine = GrdEngine(self.config)
        self._engine.sync_params(((k, v) for k, v in self.state_dict(keep_vars=True).items()), force)
        return self._engine

    def forward(self, video_feature_list, data_list, score_th=0.5, tiou_th=0.5, bins_th=0.1, nms_th=0.5, with_gt_data=True):
        """Eval contract of the reference (models/grd_model_v5.py:196-221): one video per call,
        ``data_list[0] = (uniq_quintuples (n,5), uniq_dura_inters (n,2), video_len)``; returns
        ``(pooled_se (n,k+1,2) f32, bins_probs (n,k+1) f32, bins_mask (n,k+1) bool)`` or ``(None, None)``."""
        if self.is_train:
            assert with_gt_data, "training needs the ground-truth graphs as data_list"
            return self._forward_train(video_feature_list, data_list)
        assert len(video_feature_list) == 1
        if with_gt_data:     # grounding stage evaluated alone on the GT queries (models/grd_model_v5.py:198-201)
            quint, dura, _, _ = gt_queries(data_list[0], False, int(self.PredNameEmb.shape[0]))
            video_len = data_list[0].video_len
        else:
            quint, dura, video_len = data_list[0]
        if quint is None or dura is None or quint.shape[0] == 0:
            return None, None
        out = self.forward_raw(video_feature_list[0], quint, dura, video_len, score_th, tiou_th, bins_th, nms_th)
        if int(out["status"].item()) != 0:
            # the reference raises at this point: min() of an empty selection (models/grd_model_v5.py:726)
            raise IndexError("temporal pooling selected no clip for some (query, bin): all scores are zero")
        return out["pooled_se"], out["bins_probs"], out["bins_mask"].bool()

    def _forward_train(self, video_feature_list, gt_graph_list):
        """``(total_loss, loss_dict)`` of the reference (models/grd_model_v5.py:375-389); ``total_loss.backward()`` runs the
        HIP backward and fills ``.grad`` of every trainable parameter."""
        feats = [f.to(self.EntiNameEmb.device) for f in video_feature_list]
        terms = HipGrdTrainStep.apply(self, feats, list(gt_graph_list), *self.parameters())
        return terms.sum(), {k: terms[i] for i, k in enumerate(LOSS_KEYS)}

    def forward_raw_batch(self, jobs, score_th=0.5, tiou_th=0.5, bins_th=0.1, nms_th=0.5):
        """The eval forward for several videos at once: ``jobs`` = [(video_feature (T,dim_feat), uniq_quintuples (n,5),
        uniq_dura_inters (n,2), video_len), ...] -> one dict per video (pooled_se, bins_probs, bins_mask (uint8), status), identical
        to ``forward_raw`` per video.  The reference scores one video per call (models/grd_model_v5.py:196-221); batching is what
        fills the GPU: a 200-query video is ~75 launches of a few thousand rows each."""
        with torch.no_grad():
            return self.engine().forward_batch(jobs, score_th, tiou_th, bins_th, nms_th)

    def forward_raw(self, video_feature, quint, dura, video_len, score_th, tiou_th, bins_th, nms_th, keep_intermediates=False):
        with torch.no_grad():
            return self.engine().forward(video_feature, quint, dura, video_len, score_th, tiou_th, bins_th, nms_th, keep_intermediates)
