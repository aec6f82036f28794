"""Device-resident ragged tracklet store: the packed form of a list of ``TrajProposal``.

The reference keeps one small tensor per tracklet and stretches/stacks them inside
``forward`` (models/model_0v10.py:391-430).  Here the tracklets of a batch of videos are
concatenated once into flat HBM buffers (features ``(sum L_i, D)``, boxes ``(sum L_i, 4)``)
plus int32 offset tables, which is what ``bigc_forward`` consumes (``bigc_batch`` in
``include/bigc_hip.h``).  Packing is the analogue of the reference's ``proposal.to(device)``;
a packed batch can be kept resident and reused.
"""
import numpy as np
import torch

from . import _lib


class PackedBatch(object):
    def __init__(self, proposals, device):
        """``proposals``: list of non-empty TrajProposal-like objects (see bigc_hip.synthetic.TrajProposal)."""
        assert len(proposals) > 0
        device = torch.device(device)
        self.device = device
        self.proposals = proposals
        n_per_vid = [int(p.num_proposals) for p in proposals]
        assert min(n_per_vid) > 0, "empty videos must be filtered out before packing"
        lens, tmax, wh = [], [], []
        for p in proposals:
            l = [int(b.shape[0]) for b in p.bboxes_list]
            for b, f in zip(p.bboxes_list, p.features_list):
                assert b.shape[0] == f.shape[0] and b.shape[1] == 4
            lens.append(np.asarray(l, dtype=np.int64))
            tmax.append(int(max(l)))
            wh.append([float(p.video_wh[0]), float(p.video_wh[1])])
        self.n_vid = len(proposals)
        self.n_trk = int(sum(n_per_vid))
        self.max_trk = int(max(n_per_vid))
        all_lens = np.concatenate(lens)
        self.n_rows = int(all_lens.sum())
        trk_vid = np.repeat(np.arange(self.n_vid), n_per_vid)
        t_o = (np.asarray(tmax) - 1) // 2 + 1                     # conv k3/s2/p1 output length
        self.n_conv = int((t_o[trk_vid]).sum())
        assert self.n_rows < 2 ** 31 and self.n_conv * 3 < 2 ** 31

        def on_device(t):                                         # (a no-op .to() still costs microseconds; x 2 per tracklet)
            return t if t.device == device else t.to(device)

        # every offset table in ONE host-to-device copy
        self.vid_trk_off_host = np.concatenate([[0], np.cumsum(n_per_vid)]).astype(np.int64)
        tables = [np.concatenate([[0], np.cumsum(all_lens)]), np.concatenate([[0], np.cumsum(t_o[trk_vid])]), trk_vid,
                  self.vid_trk_off_host, np.asarray(tmax)]
        sizes = [(len(a) + 3) // 4 * 4 for a in tables]          # 16-byte aligned starts
        blob = np.zeros((sum(sizes),), dtype=np.int32)
        starts = np.concatenate([[0], np.cumsum(sizes)])
        for a, o in zip(tables, starts):
            blob[o:o + len(a)] = a
        blob_d = _lib.upload(blob, device)
        self.trk_row_off, self.trk_conv_off, self.trk_vid, self.vid_trk_off, self.vid_tmax = \
            [blob_d[o:o + len(a)] for a, o in zip(tables, starts)]
        self.vid_wh = _lib.upload(np.asarray(wh, dtype=np.float32).reshape(-1, 2), device)
        self.trk_dura = torch.cat([on_device(p.traj_durations) for p in proposals], 0).to(torch.int32).contiguous()
        self.trk_cat = torch.cat([on_device(p.cat_ids) for p in proposals], 0).to(torch.int32).contiguous()
        self.trk_score = torch.cat([on_device(p.scores) for p in proposals], 0).to(torch.float32).contiguous()
        self.boxes = torch.cat([on_device(b) for p in proposals for b in p.bboxes_list], 0).to(torch.float32).contiguous()
        self.feats = torch.cat([on_device(f) for p in proposals for f in p.features_list], 0).to(torch.float32).contiguous()
        self.feat_width = int(self.feats.shape[1])
        self.feats_hr8, self.feats_hr8_flag = None, None          # see split_features()
        assert self.boxes.shape == (self.n_rows, 4) and self.feats.shape[0] == self.n_rows

    def split_features(self, dim_feat):
        """The visual feature columns once more in the split form "hr8" (pairs of f16, same footprint as f32; csrc/gemm_ps.hip): an
        eval forward then runs fc_feat2enti.0 on operands that are already split.  One pass over the features (read 4 B, write 4 B
        per element; a real ingest path writes the form while it assembles the rows, csrc/ingest.hip).  Idempotent."""
        if self.feats_hr8 is not None and self.feats_hr8.shape[1] == 4 * dim_feat:
            return
        assert dim_feat % 8 == 0 and dim_feat <= self.feat_width
        lib = _lib.load()
        self.feats_hr8 = torch.empty((self.n_rows, 4 * dim_feat), dtype=torch.uint8, device=self.device)
        self.feats_hr8_flag = torch.zeros((1,), dtype=torch.int32, device=self.device)
        _lib.check(lib.bigc_split_rows_f32(_lib.current_stream(), self.feats.data_ptr(), self.feat_width, self.feats_hr8.data_ptr(), 4 * dim_feat,
                                           self.n_rows, dim_feat, self.feats_hr8_flag.data_ptr()))

    def n_pairs(self):
        """tracklet-pairs of the batch: sum_v N_v (N_v - 1)."""
        n = np.diff(self.vid_trk_off_host)
        return int((n * (n - 1)).sum())

    def as_struct(self):
        b = _lib.bigc_batch()
        b.n_vid, b.n_trk, b.n_rows, b.n_conv, b.max_trk = self.n_vid, self.n_trk, self.n_rows, self.n_conv, self.max_trk
        for name in ("feats", "boxes", "trk_row_off", "trk_conv_off", "trk_vid", "trk_dura", "trk_cat",
                     "trk_score", "vid_trk_off", "vid_tmax", "vid_wh"):
            setattr(b, name, getattr(self, name).data_ptr())
        if self.feats_hr8 is not None:
            b.feats_hr8, b.ld_feats_hr8, b.feats_hr8_flag = self.feats_hr8.data_ptr(), int(self.feats_hr8.shape[1]), self.feats_hr8_flag.data_ptr()
        return b
