"""The explicit pairwise baseline on the HIP path: drop-in for the reference's
``models/model_pairwise_baseline.py:Base_C`` (eval forward: SURVEY.md section 8 row f3; training step :276-312).
The torch modules only own the parameters under the reference's key names; the front-end, the N x N pair fusion and the
classifier run in the HIP library (``bigc_pair_forward``); the triplet bookkeeping of ``construct_triplet`` (:306-395) is a
the HIP library too (``bigc_pair_triplets``, csrc/pair_triplet.hip: no sort, a bitmap prefix sum places every row)."""
import numpy as np
import torch
from torch import nn

from bigc_hip.engine import BigcEngine
from bigc_hip.packed import PackedBatch


class Base_C(nn.Module):
    def __init__(self, config, is_train=True):
        super().__init__()
        self.is_train = is_train
        self.config = dict(config)
        c = config
        self.num_pred_cats, self.num_enti_cats = c["num_pred_cats"], c["num_enti_cats"]
        self.dim_feat, self.dim_clsme, self.dim_enti, self.dim_ffn = c["dim_feat"], c["dim_clsme"], c["dim_enti"], c["dim_ffn"]
        self.enco_pool_len, self.use_clsme, self.rt_triplets_topk = c["enco_pool_len"], c["use_clsme"], c["rt_triplets_topk"]
        self.EntiNameEmb_path = c.get("EntiNameEmb_path", None)
        if self.EntiNameEmb_path is None:
            self.EntiNameEmb = None
        else:
            emb = torch.from_numpy(np.load(self.EntiNameEmb_path)).float()
            assert emb.shape == (self.num_enti_cats, self.dim_clsme)
            self.EntiNameEmb = nn.Parameter(emb, requires_grad=False)
        bias = torch.from_numpy(np.load(c["bias_matrix_path"])).float()
        assert bias.shape == (self.num_enti_cats, self.num_enti_cats, self.num_pred_cats)
        self.bias_matrix = nn.Parameter(bias, requires_grad=True)
        de = self.dim_enti
        self.fc_feat2enti = nn.Sequential(nn.Linear(self.dim_feat, de), nn.ReLU(), nn.Linear(de, de), nn.ReLU())
        self.fc_bbox2enti = nn.Sequential(nn.Linear(8, de), nn.ReLU(), nn.Linear(de, de), nn.ReLU())
        self.conv_feat2enti = nn.Conv1d(de * 2, de, kernel_size=3, padding=1, stride=2)
        self.fc_enti2enco = nn.Sequential(nn.Linear(de * self.enco_pool_len, de), nn.ReLU(), nn.Linear(de, de), nn.ReLU())
        k = 2 * de + (2 * self.dim_clsme if self.use_clsme else 0)
        self.fc_pred2logits = nn.Sequential(nn.Linear(k, self.dim_ffn), nn.ReLU(), nn.Linear(self.dim_ffn, self.num_pred_cats))
        for name, p in self.named_parameters():               # models/model_pairwise_baseline.py:76-91
            if name in ("bias_matrix", "EntiNameEmb"):
                continue
            if p.dim() > 1:
                nn.init.xavier_normal_(p)
            if "bias" in name:
                nn.init.zeros_(p)
        self._engine = None

    def invalidate_weights(self):
        """Call after modifying parameters through ``.data`` (in-place writes that do not bump the tensors' version counter,
        e.g. ``p.data.mul_()`` in hand-written EMA or weight-clipping code): the HIP library keeps split / re-laid-out copies of
        the weights and otherwise only refreshes them when a parameter moved or its version changed."""
        if self._engine is not None:
            self._engine.invalidate_weights()

    def engine(self, force=False):
        if self._engine is None:
            cfg = dict(self.config)
            self._engine = BigcEngine(cfg, "pairwise")
        self._engine.sync_params(((k, v) for k, v in self.state_dict(keep_vars=True).items()), force)
        return self._engine

    def forward(self, proposal_list, pos_id_list=None, label_list=None, topk=10):
        """Eval contract of the reference (:93-127): one entry per video, ``None`` for an empty video or when no pair
        overlaps in time, else ``(uniq_quintuples, uniq_scores (U,3), uniq_dura_inters, uniq_query_ids)``."""
        if self.is_train:
            assert pos_id_list is not None and label_list is not None
            return self._forward_train(proposal_list, pos_id_list, label_list)
        self.topk = topk
        keep = [i for i, p in enumerate(proposal_list) if p.num_proposals > 0]
        res = [None] * len(proposal_list)
        if not keep:
            return res
        dev = next(self.parameters()).device
        packed = PackedBatch([proposal_list[i] for i in keep], dev)
        with torch.no_grad():
            eng = self.engine()
            out = eng.pair_forward(packed)
            trip = eng.pair_triplets(packed, out, self.topk, self.rt_triplets_topk)      # construct_triplet (:314-395), all videos at once
        for j, i in enumerate(keep):
            if trip[j] is not None:
                q, s, d = trip[j]
                res[i] = (q, s, d, torch.empty(size=(s.shape[0],)))
        return res

    def _forward_train(self, proposal_list, pos_id_list, label_list):
        """(:276-312) logits of the labelled pairs of every video, BCE-with-logits mean over all of them:
        ``(total_loss, {"cls": loss})``; gradients flow to the parameters through ``HipPairTrainStep``."""
        from bigc_hip.pair_train_step import HipPairTrainStep
        assert len(proposal_list) == len(label_list) == len(pos_id_list)
        dev = next(self.parameters()).device
        packed = PackedBatch(list(proposal_list), dev)
        params = [p for _, p in self.named_parameters()]
        terms = HipPairTrainStep.apply(self, packed, list(pos_id_list), list(label_list), *params)
        loss_dict = {"cls": terms[0]}
        return torch.stack(list(loss_dict.values())).sum(), loss_dict
