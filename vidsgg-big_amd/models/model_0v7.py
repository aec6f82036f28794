"""BIG-C for VidOR on the HIP path: drop-in for the reference's ``models/model_0v7.py:BIG_C``.
Differences from the VidVRD model: fixed sine ``pos_embedding``,
classeme instead of I3D, two-layer ``fc_pred2logits``, xavier-uniform init, default topk 3."""
import numpy as np
import torch
from torch import nn

from bigc_hip.synthetic import sine_pos_emb
from ._bigc import BigcBase


class BIG_C(BigcBase):
    variant = "vidor"
    default_topk = 3           # models/model_0v7.py:362

    def _build_embeddings(self, c):
        self.use_clsme = c["use_clsme"]
        self.EntiNameEmb_path = c.get("EntiNameEmb_path", None)
        if self.EntiNameEmb_path is None:
            self.EntiNameEmb = None
        else:
            emb = torch.from_numpy(np.load(self.EntiNameEmb_path)).float()
            assert emb.shape == (self.num_enti_cats, self.dim_clsme)
            self.EntiNameEmb = nn.Parameter(emb, requires_grad=False)
        q, d = c["num_querys"], c["dim_pred"]
        self.pos_embedding = nn.Parameter(sine_pos_emb(q, d), requires_grad=False)
        self.pred_query_init = nn.Parameter(torch.zeros(q, d), requires_grad=True)

    def _build_head(self, c):
        k = self.dim_pred + 2 * self.dim_enti + (2 * self.dim_clsme if self.use_clsme else 0)
        self.fc_pred2logits = nn.Sequential(nn.Linear(k, self.dim_ffn), nn.ReLU(),
                                            nn.Linear(self.dim_ffn, self.num_pred_cats))

    def _reset_parameters(self):
        # models/model_0v7.py:346-359: xavier-uniform on matrices (biases keep torch defaults)
        for name, p in self.named_parameters():
            if name in ("bias_matrix", "EntiNameEmb", "pred_query_init"):
                continue
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        nn.init.normal_(self.pred_query_init, mean=0, std=0.1)
