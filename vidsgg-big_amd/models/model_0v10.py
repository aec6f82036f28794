"""BIG-C for VidVRD on the HIP path: drop-in for the reference's ``models/model_0v10.py:BIG_C``
(same constructor arguments, ``state_dict`` keys and ``forward`` contract)."""
import numpy as np
import torch
from torch import nn

from ._bigc import BigcBase, _mlp


class BIG_C(BigcBase):
    variant = "vidvrd"
    default_topk = 10          # models/model_0v10.py:359

    def _build_embeddings(self, c):
        self.dim_3id = c.get("dim_i3d", None)
        emb = torch.from_numpy(np.load(c["EntiNameEmb_path"])).float()
        assert emb.shape == (self.num_enti_cats, self.dim_clsme)
        self.EntiNameEmb = nn.Parameter(emb, requires_grad=False)
        q, d = c["num_querys"], c["dim_pred"]
        self.pos_embedding = nn.Parameter(torch.zeros(q, d), requires_grad=True)
        self.pred_query_init = nn.Parameter(torch.zeros(q, d), requires_grad=True)

    def _build_extra_layers(self, c):
        if self.dim_3id:
            self.fc_i3d = _mlp(self.dim_3id, self.dim_enti)

    def _build_head(self, c):
        k = self.dim_pred + 2 * self.dim_clsme + (4 if self.dim_3id else 2) * self.dim_enti
        self.fc_pred2logits = nn.Linear(k, self.num_pred_cats)

    def _reset_parameters(self):
        # same scheme as the reference (models/model_0v10.py:339-357): xavier-normal matrices,
        # zero biases, N(0, 0.1) query and position embeddings; the two loaded tables are kept
        for name, p in self.named_parameters():
            if name in ("bias_matrix", "EntiNameEmb", "pred_query_init"):
                continue
            if p.dim() > 1:
                nn.init.xavier_normal_(p)
            if "bias" in name:
                nn.init.zeros_(p)
        nn.init.normal_(self.pred_query_init, mean=0, std=0.1)
        nn.init.normal_(self.pos_embedding, mean=0, std=0.1)
